// RETIRED FROM THE LIBRARY (round 5): the all-DMA planes GEMMs (planes2, persistent planes3, stream-K) lost their A/B in the
// rollout (profiles/r03_gemm_planes3.md) and were opt-in only; kept here for the probes that include this file.
// f16x3 GEMM on fp16 OPERAND PLANES of both operands:  C = act(A W^T + bias) + R
//
// Replaces nn.Linear of the predictor blocks (reference models/Blocks/attention.py:167-175, 355-359,
// Predictors/text_cond_OCVP.py:49-50) and of the DINOv2 ViT blocks when the activation arrives as planes.
//   A: (M, 2, K) fp16 planes of 2^8 x   (hi | lo, written by the producer: LayerNorm, attention, GEMM epilogue)
//   W: (N, 2, K) fp16 planes of 2^10 w  (tocvp_split_weights_planes_f16, once per weight version)
//   product = Al Wh + Ah Wl + Ah Wh on v_mfma_f32_32x32x16_f16, fp32 accumulate: fp32-class (~2^-21).
//
// The round-1 kernels fed the weights from L2 as per-wave register fragments and split A in the k-loop:
// 48 KB crossed the CU's 64 B/clk vector-memory path per 128x128x32 tile (as long as the MFMAs took) and
// ~6 vector instructions per MFMA went into the split.  Here (geometry of the 256x256 bf16 template of
// cdna_hip_programming.md section 5, with two fp16 planes of 32 k in the place of 64 bf16 k):
//   * (64 MI) x 256 tile, 8 waves as 2 (M) x 4 (N), a wave owns (32 MI) x 64 = MI x 2 accumulator tiles;
//     MI = 4 (256 x 256: 21 B/clk of operand traffic per CU) or MI = 2 (128 x 256, more workgroups for N = 512);
//   * BOTH operands go global -> LDS by DMA (global_load_lds, 16 B per lane, no staging registers, no
//     conversion instructions); rows are 128 B = [hi k0..31 | lo k0..31]; the image is lane-linear, the
//     conflict-free order is made on the SOURCE side: physical 16-byte chunk c of row r holds logical chunk
//     c ^ ((r >> 1) & 7), so the 16 rows of every ds_read_b128 lane group hit 16 distinct slots;
//   * two LDS stages (2 x 64 KB at MI = 4), one barrier per k-tile, two waves per SIMD; the fragments of BOTH
//     k-steps of a k-tile are in registers before its barrier, so the stage is refilled right behind the barrier
//     (DMA 1.5 k-tiles ahead with two stages) and every fragment read is issued one MFMA block ahead of its use.
//
// Measured (MI355X, 38400 rows, dense random operands, scripts/gemm_shapes.py; round-1 kernel with the in-loop
// split in brackets): 2048 x 512  311 us = 259 TFLOP/s [383], 2048 x 2048  925 us = 348 [1190], 1536 x 512  227 [231],
// 512 x 2048  331 [284], 512 x 512  106 [77].  The k-loop runs at ~420 TFLOP/s; a workgroup owns its CU alone and pays
// ~24 us of prologue + epilogue without cover (its 256 KB of output leave at the ~14 B/clk/CU store-issue rate), so
// it wins only on wide or deep products.  Ablations (scripts/probes/gemm16p_ablate.hip, 2048 x 512): no DMA -14 %,
// no stores -12 %, MFMAs removed 0.63 x.  A two-workgroups-per-CU form (256 x 128 tiles, three 16-deep stages, a
// barrier per 24 MFMAs) measured 389 us on the same product, with or without a start offset that de-phases the two
// workgroups of a CU: slower, not kept.  Used for plane inputs when
// TOCVP_PRESPLIT selects them (off by default: neutral in the rollout, models/Blocks/attention.py).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

// timing experiments only (scripts/probes/gemm16p_ablate.hip): 1 = no DMA in the k-loop, 2 = no MFMAs,
// 3 = no LDS fragment reads in the k-loop, 4 = no output stores
#ifndef TOCVP_GEMM_P2_ABLATE
#define TOCVP_GEMM_P2_ABLATE 0
#endif

#ifdef TOCVP_P2_STAMP
// probe builds only (scripts/probes/gemm16p_stamp.hip): s_memtime at the phase boundaries of every workgroup
__device__ unsigned long long tocvp_p2_stamps[8192 * 4];
#define P2_STAMP(i) do { if (threadIdx.x == 0) tocvp_p2_stamps[(blockIdx.x & 8191) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P2_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int PABL = TOCVP_GEMM_P2_ABLATE;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;
constexpr int BK = 32, ROWB = 128;                  // k per stage, bytes per LDS row (2 planes x 32 k x 2 B)

struct PArgs {
    const unsigned char* A; const unsigned char* W;
    const float* bias; const float* R; int ldr;
    float* C; int ldc; int c_split;
    int M, N, K, act;
};

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TOCVP_ACT_GELU) return tocvp_gelu(v);
    return v;
}

template <int MI>
__global__ __launch_bounds__(512, 2) void gemm_f16_planes2_kernel(PArgs p) {
    constexpr int NSTAGE = 2;
    constexpr int NI = 2, BM = 64 * MI, BN = 256;
    constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * ROWB, STAGE = A_STAGE + B_STAGE;
    constexpr int A_DMA = A_STAGE / (8 * 1024), B_DMA = B_STAGE / (8 * 1024);    // 1 KiB instructions per wave
    static_assert(NSTAGE * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = p.N / BN;
    int bid = blockIdx.x;
    {   // XCD-contiguous tile order (bijective for any grid size): neighbours in the grid share A / W panels in one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int nk = p.K / BK;

    // ---- DMA source offsets: instruction i of this wave writes LDS rows (wave * CNT + i) * 8 .. + 7
    unsigned voff_a[A_DMA], voff_b[B_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);                 // logical chunk landing in this lane's slot
        const int grow = min(m0 + row, p.M - 1);
        voff_a[i] = (unsigned)((((size_t)grow * 2 + (c >> 2)) * p.K + (c & 3) * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff_b[i] = (unsigned)((((size_t)(n0 + row) * 2 + (c >> 2)) * p.K + (c & 3) * 8) * 2);
    }
    auto dma = [&](int stage, int kt) {
        const unsigned char* abase = p.A + (size_t)kt * (BK * 2);    // uniform
        const unsigned char* wbase = p.W + (size_t)kt * (BK * 2);
        unsigned char* la = lds + stage * STAGE + (wave * A_DMA) * 1024;
        unsigned char* lb = lds + stage * STAGE + A_STAGE + (wave * B_DMA) * 1024;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + voff_a[i]),
                                             (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + voff_b[i]),
                                             (__attribute__((address_space(3))) void*)(lb + i * 1024), 16, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: row = base + l31 (base a multiple of 32), chunk (plane * 4 + ks * 2 + h) ^ ((l31 >> 1) & 7)
    const int x16 = ((l31 >> 1) & 7) << 4;
    const int a_row = (wm * (32 * MI) + l31) * ROWB, b_row = A_STAGE + (wn * 64 + l31) * ROWB;
    struct Frags { f16x8 a[MI][2], b[NI][2]; };
    auto read_frags = [&](Frags& f, int stage, int ks) {
        const unsigned char* sb = lds + stage * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = (((s * 4 + (PABL == 3 ? 0 : ks) * 2 + h) << 4) ^ x16);
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[i][s] = *reinterpret_cast<const f16x8*>(sb + a_row + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < NI; ++j) f.b[j][s] = *reinterpret_cast<const f16x8*>(sb + b_row + j * 32 * ROWB + coff);
        }
    };
    auto mfma = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (PABL == 2) {
                    asm volatile("" :: "v"(f.a[i][0]), "v"(f.a[i][1]), "v"(f.b[j][0]), "v"(f.b[j][1]));
                    continue;
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][1], f.b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][0], acc[i][j], 0, 0, 0);
            }
    };

    // Software pipeline.  Both k-steps of k-tile kt sit in REGISTERS (f0, f1) before the barrier of iteration kt, so
    // its stage is refilled right behind that barrier with k-tile kt + 2 (two stages, DMA 1.5 k-tiles ahead of its
    // use), and the fragment reads of k-tile kt + 1 are issued a whole MFMA block (24 or 48 MFMAs) before their use.
    Frags f0, f1;
    P2_STAMP(0);
    // LDS-DMA completion is tracked by vmcnt only: the compiler does not wait for it at a barrier, so every wave
    // waits for its own share (asm, invisible to the waitcnt pass) and the barrier then covers everybody's.
    dma(0, 0);
    dma(1, nk > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(f0, 0, 0);
    read_frags(f1, 0, 1);
    P2_STAMP(1);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        mfma(f0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of k-tile kt + 1 (issued one iteration ago)
        __syncthreads();                               // every wave holds k-tile kt in registers; k-tile kt + 1 has landed
        if (PABL != 1) dma(kt & 1, kt + 2 < nk ? kt + 2 : nk - 1);   // past the end: a harmless re-load, never read
        read_frags(f0, (kt + 1) & 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma(f1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(f1, (kt + 1) & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    mfma(f0);
    mfma(f1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-load of the last k-tile
    __syncthreads();                                   // the epilogue reuses the stages
    P2_STAMP(2);

    // ---- epilogue: 32-row x 64-column blocks staged through LDS, written back as 16-byte rows
    constexpr int SS = 64 + 4;
    float* stage_f = reinterpret_cast<float*>(lds) + wave * (32 * SS);
    constexpr int F4R = 64 / 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage_f[acc_row(r, h) * SS + j * 32 + l31] = act_fn(acc[i][j][r] * (1.f / (SA * SW)) + bv, p.act);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < (32 * F4R) / 64; ++it) {
            const int idx = lane + 64 * it;
            const int rr = idx / F4R, c4 = (idx % F4R) * 4;
            const int row = m0 + wm * (32 * MI) + i * 32 + rr, col = n0 + wn * 64 + c4;
            if (row < p.M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stage_f + rr * SS + c4);
                if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (size_t)row * p.ldr + col);
                if (PABL == 4 && v[0] != 12345.f) continue;
                if (p.c_split)
                    tocvp_store_planes4(p.C, (size_t)row * 2 * p.N + col, (size_t)p.N, v, 22);
                else
                    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef TOCVP_P2_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P2_STAMP(3);
#endif
}

// ------------------------------------------------------------------------------------------------
// planes3 (round 3): the same tile, LDS image and arithmetic as planes2, restructured after in-kernel stamps
// (scripts/probes/gemm16p_stamp.hip, 38400 x 2048 x 512: prologue 8.4k, k-loop 71.4k = 4460 cycles per k-tile
// against 3072 of matrix time, epilogue 27.5k cycles per 256 x 256 tile) showed where planes2 loses:
//   * its k-tile is four PHASES (24 MFMAs | wait + barrier | 8 DMA + 12 fragment reads | 24 MFMAs | 12 reads) that
//     both waves of a SIMD run in lockstep, so the matrix pipe idles through every issue burst.  Here the reads and
//     the DMA instructions are woven BETWEEN the MFMAs (sched_group_barrier): block A = 24 MFMAs of k-step 0 with
//     the 12 reads of k-step 1, block B = 24 MFMAs of k-step 1 with the 8 DMA instructions of k-tile g + 2 and the
//     12 reads of k-tile g + 1 / k-step 0; one barrier per k-tile as before;
//   * PERSISTENT workgroups (one per CU) walk their tiles as ONE stream of k-tiles: the DMA runs two k-tiles ahead
//     across tile boundaries, so a tile's prologue (first DMA + its latency) hides under the previous tile's last
//     MFMAs and its epilogue;
//   * the MFMA operands are swapped (D^T = W A^T): a lane then holds FOUR CONSECUTIVE COLUMNS of one output row per
//     register quad, so the epilogue stores 16 bytes per lane straight from the accumulators -- no LDS staging
//     (the stages already hold the next tile), no second pass.
// ------------------------------------------------------------------------------------------------
__device__ float tocvp_p3_zero_bias[4];                          // bias = nullptr reads these zeros (no branch in the epilogue)

#ifdef TOCVP_P3_STAMP
__device__ unsigned long long tocvp_p3_stamps[256 * 4];      // per workgroup: start, end, sum of epilogue cycles, tiles
#endif

// Stream-K hand-off of a cut tile (rare paths: at most once each per workgroup and launch).  They are separate,
// NOT inlined functions working on a copy of the accumulators in private memory: inlined, their address arithmetic
// and the 128 live accumulators pushed the k-tile loop over the register budget (fragments and accumulators spilled
// in every iteration: 3x slower).  Protocol = cdna_hip_programming.md Guideline 16: plain stores -> every storing
// wave drains -> barrier -> one lane: agent-scope release, drain, relaxed agent flag store;  consumer: one lane polls
// the flag relaxed, agent-scope acquire, drain -> barrier -> plain loads; the flag is re-armed (zero) by the consumer.
template <int NT>
__device__ __noinline__ void p3_park(const f32x16* tmp, float* ws_part, unsigned* ws_flag, int slot) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* mine = ws_part + ((size_t)slot * 8 + wave) * (NT * 16 * 64) + lane * 4;
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(mine + (k * 4 + q) * 256) =
                f32x4{tmp[k][4 * q], tmp[k][4 * q + 1], tmp[k][4 * q + 2], tmp[k][4 * q + 3]};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(ws_flag + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NT>
__device__ __noinline__ void p3_gather(f32x16* tmp, const float* ws_part, unsigned* ws_flag, int w) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) {
        while (__hip_atomic_load(ws_flag + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
            __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float* theirs = ws_part + ((size_t)w * 8 + wave) * (NT * 16 * 64) + lane * 4;
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + (k * 4 + q) * 256);
#pragma unroll
            for (int u = 0; u < 4; ++u) tmp[k][4 * q + u] += v[u];
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave has read the record: the flag may be re-armed
    if (t == 0) __hip_atomic_store(ws_flag + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MI, bool SK>
__global__ __launch_bounds__(512, 2) void gemm_f16_planes3_kernel(PArgs p, int ntm, int ntn, int skew, float* ws_part,
                                                               unsigned* ws_flag) {
    constexpr int NI = 2, BM = 64 * MI, BN = 256;
    constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * ROWB, STAGE = A_STAGE + B_STAGE;
    constexpr int A_DMA = A_STAGE / (8 * 1024), B_DMA = B_STAGE / (8 * 1024);    // 1 KiB instructions per wave
    static_assert(2 * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = ntm * ntn, G = gridDim.x;
    // workgroups are dealt round-robin over the 8 XCDs: those sharing an XCD (same blockIdx & 7) take NEIGHBOURING
    // tiles of every round (row-block major: they share A row panels and all of W in that XCD's L2)
    int slot = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, xcd = slot & 7;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (slot >> 3);
    }
    const int nk = p.K / BK;
    // The stream of this workgroup: `total` k-tiles starting at k-tile kt0 of tile `first_tile`, tile after tile
    // with stride `tstride`.
    //   data-parallel (ws_part == nullptr): whole tiles slot, slot + G, ...;
    //   stream-K (ws_part != nullptr): the ntiles * nk k-tiles of the product are cut into G EQUAL contiguous
    //   ranges, so no CU idles in a last, partly filled round (300 tiles of a 512-wide product on 256 CUs: 2 rounds at
    //   59 %).  A tile cut between workgroups is finished by the one holding its FIRST k-range: it reaches that tile
    //   at the END of its stream, when the holders of the later k-ranges (which meet the tile at the START of theirs)
    //   have long written their partial accumulators to the workspace; it adds them in k order (deterministic).
    int first_tile, kt0, total, tstride;
    long u0 = 0, u1 = 0;
    if (SK) {
        const long U = (long)ntiles * nk;
        u0 = U * slot / G;
        u1 = U * (slot + 1) / G;
        first_tile = (int)(u0 / nk);
        kt0 = (int)(u0 - (long)first_tile * nk);
        total = (int)(u1 - u0);
        tstride = 1;
    } else {
        first_tile = slot;
        kt0 = 0;
        total = slot < ntiles ? ((ntiles - slot + G - 1) / G) * nk : 0;
        tstride = G;
    }
    if (total == 0) return;
#ifdef TOCVP_P3_STAMP
    unsigned long long st_epi = 0, st_start = __builtin_amdgcn_s_memtime();
#endif
    if (skew > 0) {          // de-phase the epilogue bursts of the workgroups (experiment)
        const int ph = (blockIdx.x >> 3) & 3;
        for (int i = 0; i < ph * skew; ++i) __builtin_amdgcn_s_sleep(64);
    }

    // ---- DMA: per-lane offsets inside a tile (32 bit), uniform tile / k-tile base
    const size_t row_bytes = (size_t)p.K * 4;                        // one row of planes: 2 planes x K x 2 B
    unsigned voff_a[A_DMA], voff_b[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);                 // logical 16-byte chunk landing in this lane's slot
        voff_b[i] = (unsigned)(((size_t)row * 2 + (c >> 2)) * p.K * 2 + (c & 3) * 16);
    }
    auto set_voff_a = [&](int m0) {
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) {
            const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int rowc = min(row, p.M - 1 - m0);                 // rows past M re-read the last row (never stored)
            voff_a[i] = (unsigned)(((size_t)rowc * 2 + (c >> 2)) * p.K * 2 + (c & 3) * 16);
        }
    };
    // DMA cursor: (tile, k-tile) of the NEXT k-tile to fetch; runs two k-tiles ahead of the MFMAs
    int d_tile = first_tile, d_kt = kt0, d_left = total;
    const unsigned char* d_abase = p.A + (size_t)((d_tile / ntn) * BM) * row_bytes;
    const unsigned char* d_wbase = p.W + (size_t)((d_tile % ntn) * BN) * row_bytes;
    set_voff_a((d_tile / ntn) * BM);
    // one k-tile = A_DMA + B_DMA instructions per wave.  The A pieces are issued in block B of iteration g (k-tile
    // g + 2), the W pieces of the same k-tile in block A of iteration g + 1 (from the cursor saved before it
    // advanced): at most 4 DMA instructions per 24 MFMAs, one every six.  An LDS-DMA instruction occupies the CU's
    // address path for ~16 cycles and blocks its wave while that path is busy; 8 per wave right behind the barrier
    // in all 8 waves starved the matrix pipe (ablation: 730 of 3980 cycles per k-tile).
    auto dma_a = [&](int stage) {
        const unsigned char* ab = d_abase + (size_t)d_kt * (BK * 2);   // uniform
        unsigned char* la = lds + stage * STAGE + (wave * A_DMA) * 1024;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + voff_a[i]),
                                             (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
    };
    auto dma_w = [&](int stage, const unsigned char* wb) {
        unsigned char* lb = lds + stage * STAGE + A_STAGE + (wave * B_DMA) * 1024;
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + voff_b[i]),
                                             (__attribute__((address_space(3))) void*)(lb + i * 1024), 16, 0, 0);
    };
    auto w_now = [&]() { return d_wbase + (size_t)d_kt * (BK * 2); };
    auto dma_advance = [&]() {       // past the end the cursor stays on the last k-tile: a harmless re-load, never read
        if (d_left > 1) {
            --d_left;
            if (++d_kt == nk) {
                d_kt = 0;
                d_tile += tstride;
                d_abase = p.A + (size_t)((d_tile / ntn) * BM) * row_bytes;
                d_wbase = p.W + (size_t)((d_tile % ntn) * BN) * row_bytes;
                set_voff_a((d_tile / ntn) * BM);
            }
        }
    };

    // bias of a column quad: p.bias + column, or four zeros when there is no bias (index scaled by 0)
    const float* bias_or_zero = p.bias ? p.bias : tocvp_p3_zero_bias;
    const int bias_step = p.bias ? 1 : 0;
    f32x16 acc[MI][NI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    const int x16 = ((l31 >> 1) & 7) << 4;
    const int a_row = (wm * (32 * MI) + l31) * ROWB, b_row = A_STAGE + (wn * 64 + l31) * ROWB;
    struct Frags { f16x8 a[MI][2], b[NI][2]; };
    auto read_frags = [&](Frags& f, int stage, int ks) {
        const unsigned char* sb = lds + stage * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = (((s * 4 + ks * 2 + h) << 4) ^ x16);
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[i][s] = *reinterpret_cast<const f16x8*>(sb + a_row + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < NI; ++j) f.b[j][s] = *reinterpret_cast<const f16x8*>(sb + b_row + j * 32 * ROWB + coff);
        }
    };
    // D^T tile: rows (accumulator registers) = 32 columns n of W, columns (lanes) = 32 rows m of A
    auto mfma = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j][0], f.a[i][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j][1], f.a[i][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j][0], f.a[i][0], acc[i][j], 0, 0, 0);
            }
    };

    // ---- epilogue of one tile, straight from the accumulators: register quad q of tile (i, j) holds columns
    // n0 + wn*64 + 32 j + 8 q + 4 h .. + 3 of row m0 + wm*32*MI + 32 i + l31.  Branch-free inside: the activation,
    // residual and output-format choices are compile-time parameters of the body (one uniform switch per tile).
    const float act_floor = p.act == TOCVP_ACT_RELU ? 0.f : -3.0e38f;      // none / ReLU as one v_max
    auto epilogue_body = [&](int tile, auto gelu_c, auto has_r_c, auto csplit_c) __attribute__((always_inline)) {
        constexpr bool GELU = decltype(gelu_c)::value, HASR = decltype(has_r_c)::value, CSPLIT = decltype(csplit_c)::value;
        const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
        const int nb = n0 + wn * 64 + 4 * h;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bias_or_zero + bias_step * (nb + 32 * j + 8 * q));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = m0 + wm * (32 * MI) + 32 * i + l31;
                if (row < p.M) {
                    f32x4 rq[4];
                    if (HASR) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            rq[q] = *reinterpret_cast<const f32x4*>(p.R + (size_t)row * p.ldr + nb + 32 * j + 8 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = nb + 32 * j + 8 * q;
                        f32x4 v;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float x = acc[i][j][4 * q + u] * (1.f / (SA * SW)) + bq[q][u];
                            v[u] = GELU ? tocvp_gelu(x) : fmaxf(x, act_floor);
                        }
                        if (HASR) v += rq[q];
#if defined(TOCVP_P3_STORE) && TOCVP_P3_STORE == 0
                        if (v[0] == 12345.f && v[1] == 54321.f)            // timing experiment: no stores
#endif
                        if (CSPLIT)
                            tocvp_store_planes4(p.C, (size_t)row * 2 * p.N + col, (size_t)p.N, v, 22);
                        else
                            *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
                    }
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    const int epi_sel = (p.act == TOCVP_ACT_GELU ? 4 : 0) | (p.R ? 2 : 0) | (p.c_split ? 1 : 0);
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        switch (epi_sel) {
            case 0: epilogue_body(tile, F_{}, F_{}, F_{}); break;
            case 1: epilogue_body(tile, F_{}, F_{}, T_{}); break;
            case 2: epilogue_body(tile, F_{}, T_{}, F_{}); break;
            case 3: epilogue_body(tile, F_{}, T_{}, T_{}); break;
            case 4: epilogue_body(tile, T_{}, F_{}, F_{}); break;
            case 5: epilogue_body(tile, T_{}, F_{}, T_{}); break;
            case 6: epilogue_body(tile, T_{}, T_{}, F_{}); break;
            default: epilogue_body(tile, T_{}, T_{}, T_{}); break;
        }
    };

    // ---- the stream of k-tiles.  Stage g & 1 holds k-tile g.  At the top of iteration g: F0 = k-step 0 of k-tile g
    // (in flight or landed), k-tile g + 1 is on its way into the other stage.
    Frags F0, F1;
    dma_a(0);
    dma_w(0, w_now());
    dma_advance();
    dma_a(1);
    const unsigned char* w_late = w_now();             // W pieces of k-tile 1: issued in block A of iteration 0
    dma_advance();
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_DMA) : "memory");      // k-tile 0 landed (k-tile 1's A pieces may fly)
    __syncthreads();
    read_frags(F0, 0, 0);
    int c_tile = first_tile, c_kt = kt0, c_ka = kt0;      // tile in the accumulators, next k-tile, first k-tile held
    constexpr int NRD = 2 * (MI + NI);                 // fragment reads per k-step
    // weave<ND>(): the block's 3 MI NI MFMAs with its ND DMA instructions and NRD fragment reads between them
    // (sched_group_barrier pipeline; memory instructions spread evenly, a DMA first, then reads)
    auto weave_impl = [](auto nd_c) __attribute__((always_inline)) {
        constexpr int ND = decltype(nd_c)::value, NMEM = ND + NRD, NM = 3 * MI * NI;
        constexpr int PER = NM >= NMEM ? NM / NMEM : 1;                // MFMAs in front of every memory instruction
#pragma unroll
        for (int i = 0; i < NMEM; ++i) {
            if (i * PER < NM) __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                 // MFMA
            // DMA instructions in the first half of the block (they have to land before the next barrier)
            if (i < 2 * ND && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // LDS DMA
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                // DS read
        }
    };
    for (int g = 0; g < total; ++g) {
        const int stage = g & 1;
        // block A: k-step 0 products; woven in: the reads of k-step 1 and the W pieces of k-tile g + 1
        dma_w(stage ^ 1, w_late);
        read_frags(F1, stage, 1);
        mfma(F0);
        weave_impl(std::integral_constant<int, B_DMA>{});
        __builtin_amdgcn_sched_barrier(0);
        // every wave has read all of this stage (F1 landed), and its share of k-tile g + 1 has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        // block B: k-step 1 products; woven in: the A pieces of k-tile g + 2 (into the stage just released) and the
        // reads of k-tile g + 1 / k-step 0
#ifndef TOCVP_P3_NODMA
        dma_a(stage);
#endif
        w_late = w_now();
        read_frags(F0, stage ^ 1, 0);
        mfma(F1);
        weave_impl(std::integral_constant<int, A_DMA>{});
        __builtin_amdgcn_sched_barrier(0);
        dma_advance();
        if (++c_kt == nk || (SK && g == total - 1)) {
#ifdef TOCVP_P3_STAMP
            const unsigned long long e0 = __builtin_amdgcn_s_memtime();
#endif
            if (!SK || (c_ka == 0 && c_kt == nk)) {
                epilogue(c_tile);                      // the whole k range was accumulated here
            } else if (c_ka > 0) {
                // a LATER k-range of a tile that starts in another workgroup's stream: park the partial sums
                f32x16 tmp[MI * NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) tmp[i * NI + j] = acc[i][j];
                p3_park<MI * NI>(tmp, ws_part, ws_flag, slot);
            } else {
                // the FIRST k-range of a tile whose later ranges live in the following workgroups' streams (they met
                // this tile first): add their partial sums in k order, then finish the tile
                f32x16 tmp[MI * NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) tmp[i * NI + j] = acc[i][j];
                int covered = c_kt;                    // k-tiles of this tile accumulated so far
                for (int w = slot + 1; covered < nk; ++w) {
                    const long U = (long)ntiles * nk;
                    const long w0 = U * w / G, w1 = U * (w + 1) / G;
                    const long tile_end = ((long)c_tile + 1) * nk;
                    covered += (int)((w1 < tile_end ? w1 : tile_end) - w0);
                    p3_gather<MI * NI>(tmp, ws_part, ws_flag, w);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = tmp[i * NI + j];
                epilogue(c_tile);
            }
            zero_acc();
#ifdef TOCVP_P3_STAMP
            st_epi += __builtin_amdgcn_s_memtime() - e0;
#endif
            c_kt = 0;
            c_ka = 0;
            c_tile += tstride;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the tail
#ifdef TOCVP_P3_STAMP
    if (t == 0) {
        tocvp_p3_stamps[blockIdx.x * 4 + 0] = st_start;
        tocvp_p3_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        tocvp_p3_stamps[blockIdx.x * 4 + 2] = st_epi;
        tocvp_p3_stamps[blockIdx.x * 4 + 3] = (unsigned long long)((total + nk - 1) / nk);
    }
#endif
}

// W (N, K) fp32 -> (N, 2, K) fp16 planes of 2^10 w
__global__ __launch_bounds__(256) void split_weights_planes_f16_kernel(const float* __restrict__ w,
                                                                       _Float16* __restrict__ out, long n, int K) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = i / K;
    const int k = (int)(i - row * K);
    const float X = fminf(fmaxf(w[i] * SW, -65504.f), 65504.f);
    const _Float16 hi = (_Float16)X;
    out[(row * 2 + 0) * K + k] = hi;
    out[(row * 2 + 1) * K + k] = (_Float16)(X - (float)hi);
}

}  // namespace

extern "C" int tocvp_split_weights_planes_f16(const float* w, void* out, int N, int K, void* stream) {
    TOCVP_CHECK_ARG(w && out && N > 0 && K > 0);
    const long n = (long)N * K;
    hipLaunchKernelGGL(split_weights_planes_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(out), n, K);
    return tocvp_launch_status();
}

namespace {
constexpr size_t P3_FLAG_BYTES = 4096;                 // one flag word per workgroup, in front of the records
constexpr size_t P3_REC_BYTES = 8 * 64 * 64 * 4;       // accumulators of one 128 x 256 tile: 8 waves x 64 registers

int p3_cus() {
    static const int ncu = []() {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        const char* e = getenv("TOCVP_GEMM_P3_CUS");
        return std::max(1, std::min(e ? atoi(e) : n, (int)(P3_FLAG_BYTES / 4)));
    }();
    return ncu;
}
}  // namespace

extern "C" size_t tocvp_gemm_f16planes_ws_bytes(void) { return P3_FLAG_BYTES + (size_t)p3_cus() * P3_REC_BYTES; }

extern "C" int tocvp_gemm_f16planes_ws_f32(const void* A_planes, const void* W_planes, const float* bias,
                                           const float* R, int ldr, void* C, int c_split, int ldc, int M, int N,
                                           int K, int act, void* ws, size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(A_planes && W_planes && C);
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 32) == 0 && (N % 256) == 0);
    TOCVP_CHECK_ARG(c_split || (ldc >= N && (ldc & 3) == 0 && tocvp_aligned16(C)));
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= N && (ldr & 3) == 0 && tocvp_aligned16(R)));
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    TOCVP_CHECK_ARG((size_t)N * 2 * K * 2 < 0xffffffffull);                   // 32-bit offsets inside the W image
    TOCVP_CHECK_ARG(ws == nullptr || (ws_bytes >= tocvp_gemm_f16planes_ws_bytes() && tocvp_aligned16(ws)));
    if (!tocvp_aligned16(A_planes) || !tocvp_aligned16(W_planes)) return TOCVP_EALIGN;
    if (bias && !tocvp_aligned16(bias)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    PArgs p{static_cast<const unsigned char*>(A_planes), static_cast<const unsigned char*>(W_planes), bias, R, ldr,
            static_cast<float*>(C), ldc, c_split ? 1 : 0, M, N, K, act};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ntn = N / 256, nk = K / 32;
    static const int use_p3 = []() { const char* e = getenv("TOCVP_GEMM_P3"); return e ? atoi(e) : 1; }();
    static const int force = []() { const char* e = getenv("TOCVP_GEMM_P2_MI"); return e ? atoi(e) : 0; }();
    static const int force_sk = []() { const char* e = getenv("TOCVP_GEMM_P3_SK"); return e ? atoi(e) : -1; }();
    if (use_p3) {
        // Persistent planes3, one workgroup per CU, whole tiles ("data-parallel").  The rows are cut into PHASES of
        // decreasing tile height so that no phase ends in a nearly empty round of CUs: as many FULL rounds of
        // 256-row tiles as the CU count allows, then of 128-row tiles, and the rest as one round of 64-row tiles
        // (38400 x 512: 256 + 176 tiles instead of 300 tiles in two rounds at 59 %).  Each phase is one launch on
        // its own row range (row-offset pointers): no cross-workgroup reduction, results independent of the cut.
        // Cost model in cycles per workgroup (measured, scripts/probes/gemm16p3_check.hip): a k-tile of
        // 256 / 128 / 64 x 256 x 32 ~3900 / 2600 / 1700 cycles, an epilogue ~18000 / 9500 / 5500.
        // Stream-K on 128-row tiles (needs the workspace) is kept behind TOCVP_GEMM_P3_SK=1: measured slower than
        // the phases (354 vs ~200 us on 38400 x 512 x 2048; the 256-row kernel has no register left for its
        // hand-off code, and 128-row tiles run the k-loop at 59 % of the matrix rate).
        const int ncu = p3_cus();
        static const int skew = []() { const char* e = getenv("TOCVP_GEMM_P3_SKEW"); return e ? atoi(e) : 0; }();
        auto launch = [&](int mi, int m_begin, int rows, bool sk) {
            PArgs q = p;
            q.A = p.A + (size_t)m_begin * K * 4;
            q.C = c_split ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(p.C) + (size_t)m_begin * 2 * N)
                          : p.C + (size_t)m_begin * ldc;
            q.R = p.R ? p.R + (size_t)m_begin * ldr : nullptr;
            q.M = rows;
            const int ntm = (rows + 64 * mi - 1) / (64 * mi);
            const unsigned grid = sk ? (unsigned)ncu : (unsigned)std::min<long>((long)ntm * ntn, ncu);
            float* part = sk ? reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + P3_FLAG_BYTES) : nullptr;
            unsigned* flag = sk ? static_cast<unsigned*>(ws) : nullptr;
            if (sk)
                hipLaunchKernelGGL((gemm_f16_planes3_kernel<2, true>), dim3(grid), dim3(512), 0, s, q, ntm, ntn, skew, part, flag);
            else if (mi == 4)
                hipLaunchKernelGGL((gemm_f16_planes3_kernel<4, false>), dim3(grid), dim3(512), 0, s, q, ntm, ntn, skew, part, flag);
            else if (mi == 2)
                hipLaunchKernelGGL((gemm_f16_planes3_kernel<2, false>), dim3(grid), dim3(512), 0, s, q, ntm, ntn, skew, part, flag);
            else
                hipLaunchKernelGGL((gemm_f16_planes3_kernel<1, false>), dim3(grid), dim3(512), 0, s, q, ntm, ntn, skew, part, flag);
        };
        if (force_sk == 1 && ws && (long)((M + 127) / 128) * ntn * nk >= ncu) {
            launch(2, 0, M, true);
            return tocvp_launch_status();
        }
        if (force) {
            launch(force, 0, M, false);
            return tocvp_launch_status();
        }
        static const double KT[5] = {0, 1700., 2600., 0, 3900.}, EPI[5] = {0, 5500., 9500., 0, 18000.};
        const double epi_scale = c_split ? 1.7 : 1.0;
        auto tile_cost = [&](int mi) { return nk * KT[mi] + EPI[mi] * epi_scale; };
        // cost of finishing `rows` rows in ceil-rounds of height-mi tiles
        auto tail_cost = [&](int mi, int rows) {
            const long tiles = (long)((rows + 64 * mi - 1) / (64 * mi)) * ntn;
            return (double)((tiles + ncu - 1) / ncu) * tile_cost(mi);
        };
        int m_begin = 0, left = M;
        for (int mi : {4, 2, 1}) {
            if (left <= 0) break;
            const int h = 64 * mi;
            // best way to finish from here with this height alone
            double finish = tail_cost(mi, left);
            bool finish_here = true;
            if (mi > 1) {
                // alternative: only the full rounds at this height, the remainder at smaller heights (estimated with
                // the next height down finishing everything that is left)
                const long panels_full = ((long)ncu * (((long)(left / h) * ntn) / ncu)) / ntn;     // whole panels in full rounds
                const int rows_full = (int)panels_full * h;
                if (rows_full > 0 && rows_full < left) {
                    const double alt = (double)((panels_full * ntn + ncu - 1) / ncu) * tile_cost(mi) +
                                       std::min(tail_cost(mi / 2, left - rows_full),
                                                mi == 4 ? tail_cost(1, left - rows_full) : 1e300);
                    if (alt < finish) {
                        launch(mi, m_begin, rows_full, false);
                        m_begin += rows_full;
                        left -= rows_full;
                        finish_here = false;
                    }
                } else if (rows_full == 0) {
                    // not even one full round at this height: smaller tiles may spread the rows over more CUs
                    if (std::min(tail_cost(mi / 2, left), mi == 4 ? tail_cost(1, left) : 1e300) < finish) finish_here = false;
                }
            }
            if (finish_here) {
                launch(mi, m_begin, left, false);
                left = 0;
            }
        }
        return tocvp_launch_status();
    }
    TOCVP_CHECK_ARG((size_t)M * 2 * K * 2 < 0xffffffffull);                   // planes2: 32-bit DMA offsets
    // 256-row tiles when they still fill the chip about twice over, else 128-row tiles
    const long big = (long)((M + 255) / 256) * ntn;
    const int mi = force ? force : (big >= 448 ? 4 : 2);
    if (mi == 4)
        hipLaunchKernelGGL(gemm_f16_planes2_kernel<4>, dim3((unsigned)(((M + 255) / 256) * ntn)), dim3(512), 0, s, p);
    else
        hipLaunchKernelGGL(gemm_f16_planes2_kernel<2>, dim3((unsigned)(((M + 127) / 128) * ntn)), dim3(512), 0, s, p);
    return tocvp_launch_status();
}

extern "C" int tocvp_gemm_f16planes_f32(const void* A_planes, const void* W_planes, const float* bias,
                                        const float* R, int ldr, void* C, int c_split, int ldc, int M, int N,
                                        int K, int act, void* stream) {
    return tocvp_gemm_f16planes_ws_f32(A_planes, W_planes, bias, R, ldr, C, c_split, ldc, M, N, K, act, nullptr, 0,
                                       stream);
}
