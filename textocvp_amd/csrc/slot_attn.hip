// One slot-attention iteration over the H*W feature grid (reference: models/Blocks/attention.py:99-103).
//
//   dots[i,j] = q_i . k_j * scale ; attn = softmax_i(dots) + eps ; upd_i = sum_j attn_ij v_j / sum_j attn_ij
//
// HBM-bound: k and v (N x D fp32 each, 4 MiB per sample at 64x64) are streamed exactly once per
// iteration with coalesced 16-byte loads of whole 512-byte rows; everything else lives on chip.
// ONE launch (the cross-workgroup reduction is done by the last-arriving workgroup of a sample).
//
// gfx950 mapping (one 4-wave workgroup per CU, one wave per SIMD, 512 registers per lane):
//  * a wave walks 32-location tiles; the NEXT tile's k and v rows are already in flight in registers
//    (2 x 16 KiB per wave, 128 KiB per CU in flight: enough to cover the HBM latency at full rate)
//    while the current one is multiplied;
//  * both contractions run on the f16 matrix cores with split operands (X = 2^8 x = Xh + Xl in fp16
//    planes, product = Xl Yh + Xh Yl + Xh Yh, fp32 accumulate: fp32-class, ~2^-21 per product): the exact
//    fp32 MFMA (64 cycles per K = 2) made this kernel matrix-bound at ~1.8 TB/s, the split form needs
//    5.3x fewer matrix cycles and leaves HBM as the limit.  Valid for |q*scale|, |k|, |v| < 255;
//  * dots = q k^T is computed with the SLOTS ON THE ACCUMULATOR REGISTERS (rows) and the locations on
//    the lanes: the softmax over slots is 15 in-lane max / add steps plus ONE cross-half exchange per
//    tile (slots on lanes cost ten 32-lane butterflies per register);
//  * the tile's k and v fp16 planes live in wave-private LDS images (no workgroup barrier in the loop;
//    the attn tile re-uses the k image once the dots are done); attn and v are consumed as MFMA operands
//    through ds_read_b64_tr_b16 (hardware transpose: the contraction runs over the location index,
//    which sits on the lanes of attn).  Both planes of tile t are written at the top of the iteration
//    and the loads of tile t + 1 issued right behind them, a whole iteration ahead of their use;
//  * per-workgroup partial sums are reduced over the 4 waves in LDS, written once (16.5 KB per
//    workgroup), and the workgroup that draws the last ticket of its sample adds the records in a
//    FIXED order (deterministic) and renormalises.  Hand-off: plain stores -> every wave drains ->
//    barrier -> agent-scope release -> ticket (relaxed agent atomic) -> last arriver: agent-scope
//    acquire -> barrier -> plain loads (cdna_hip_programming.md, Guideline 16).  The ticket words are
//    zeroed by a memset node ahead of every launch.
//
// Two input forms of k / v:
//  * PLANES = false: fp32 rows (tocvp_slot_attn_iter_f32); the wave splits them into fp16 planes itself
//    (~6 vector instructions per element: the kernel is then issue-bound at ~4.5 TB/s);
//  * PLANES = true: the fp16 planes as the fused [to_k; to_v] projection GEMM wrote them
//    (tocvp_slot_attn_iter_planes_f32): per location one 1 KiB row [k hi | v hi | k lo | v lo] of
//    2^8 * value -- the same HBM bytes as fp32, no conversion work left in the loop (one 16-byte load and
//    one ds_write_b128 per 8 elements), and the three iterations of the first frame share the split.
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int SD = 128;                       // slot / feature dim handled by this kernel
constexpr int KROW = 272;                     // bytes per k-plane row  (256 + 16: ds_read_b128 rows on distinct slots)
constexpr int VROW = 320;                     // bytes per v-plane row  (256 + 64: 4 transposed-read rows tile the 64 banks)
constexpr int KPLANE = 32 * KROW;             // one fp16 plane of a 32-location k tile
constexpr int VPLANE = 32 * VROW;
constexpr int AROW = 64;                      // attn image row: 32 slots x fp16 (lives in the k image after the dots)
constexpr int APLANE = 32 * AROW;
constexpr int WAVE_LDS = 2 * KPLANE + 2 * VPLANE;     // 37888 B per wave, 151552 B per workgroup
constexpr int REC = SD * 32 + 32;             // floats per partial record: upd [slot][d] + rowsum[slot]
constexpr int CNT_BYTES = 1024;               // ticket words in front of the records (B <= 65535 -> sized per call)
constexpr float SA = TOCVP_F16X3_ACT_SCALE;   // 2^8
constexpr float F16MAX = 65504.f;
constexpr float NEG_BIG = -1.0e30f;

struct SaArgs {
    const float* q; const float* k; const float* v; int ldkv;   // PLANES: k = the plane rows, v unused, ldkv = 256
    float* attn_out; float* rec; unsigned* cnt; float* updates;
    int B, Ks, N; int tpw, nrec;          // tiles per wave, workgroups (= records) per sample
    float scale, eps;
};

__device__ __forceinline__ float clampf(float v, float m) { return __builtin_amdgcn_fmed3f(v, -m, m); }

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// four fp32 -> fp16 planes of 2^8 x (saturating)
__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float X = clampf(v[u] * SA, F16MAX);
        hi[u] = (_Float16)X;
        lo[u] = (_Float16)(X - (float)hi[u]);
    }
}

// MFMA 32x32x16 operand fragment (8 consecutive k for this lane's row / column) out of a ROW-MAJOR
// [k][column] fp16 image, by two hardware-transposed reads: a 16-lane group reads a block of 4 k-rows x 16
// columns; lane 4q+p of the group supplies the address of row q, columns 4p..4p+3, lane i receives column i
// of the 4 rows (cdna_hip_programming.md, T10).  ``addr`` = this lane's address for k-rows 0..3 of the step.
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* addr, int row_stride) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(addr));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(addr + 4 * row_stride));
    union { s16x4 s[2]; f16x8 f; } u;
    u.s[0] = a;
    u.s[1] = b;
    return u.f;
}

template <bool PLANES>
__global__ __launch_bounds__(256, 1) void slot_attn_kernel(SaArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVE_LDS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // provably wave-uniform: scalar loop control
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, g = blockIdx.x;
    unsigned char* wl = lds + wave * WAVE_LDS;            // [k hi | k lo] (attn hi | lo over it later) [v hi | v lo]
    unsigned char* wv = wl + 2 * KPLANE;

    const int ntiles = p.N / 32;
    const int first = (g * 4 + wave) * p.tpw;
    const int last = min(first + p.tpw, ntiles);
    const char* kb = reinterpret_cast<const char*>(p.k + (size_t)b * p.N * p.ldkv);     // planes: 1 KiB rows (ldkv 256)
    const char* vb = PLANES ? kb : reinterpret_cast<const char*>(p.v + (size_t)b * p.N * p.ldkv);
    const unsigned row_bytes = (unsigned)p.ldkv * 4u;

    // tile loader.  fp32 rows: instruction `it` covers rows 2 it, 2 it + 1 (lane: row 2 it + (lane >> 5), 16 B at
    // column 4 l31) of k (kt) and of v (vt).  Planes: instruction `it` covers the whole 1 KiB row `it` = [k hi |
    // v hi | k lo | v lo] (lane: 16 B = 8 fp16 of the image lane >> 4), rows 0..15 in kt, 16..31 in vt.
    f32x4 kt[16], vt[16];
    auto load_tile = [&](const char* base, int tile, f32x4 (&dst)[16]) {
        const char* src = base + (size_t)tile * 32 * row_bytes + (unsigned)h * row_bytes + (unsigned)l31 * 16u;
#pragma unroll
        for (int it = 0; it < 16; ++it)
            dst[it] = *reinterpret_cast<const f32x4*>(src + (unsigned)(2 * it) * row_bytes);
    };
    auto load_planes = [&](int tile) {
        const char* src = kb + (size_t)tile * 32 * 1024 + (unsigned)lane * 16u;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            kt[it] = *reinterpret_cast<const f32x4*>(src + it * 1024);
            vt[it] = *reinterpret_cast<const f32x4*>(src + (16 + it) * 1024);
        }
    };
    if (first < last) {
        if (PLANES) {
            load_planes(first);
        } else {
            load_tile(kb, first, kt);
            load_tile(vb, first, vt);
        }
    }

    // ---- q operand (A of dots = q k^T): row = slot l31, k = d 16 ks + 8 h + j; softmax scale folded in
    f16x8 qh[8], ql[8];
    {
        const bool ok = l31 < p.Ks;
        const float* qrow = p.q + ((size_t)b * p.Ks + (ok ? l31 : 0)) * SD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f32x4 a0 = *reinterpret_cast<const f32x4*>(qrow + 16 * ks);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(qrow + 16 * ks + 4);
            f16x4 h0, l0, h1, l1;
            split4(ok ? a0 * p.scale : a0 * 0.f, h0, l0);
            split4(ok ? a1 * p.scale : a1 * 0.f, h1, l1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                qh[ks][u] = h0[u]; qh[ks][4 + u] = h1[u];
                ql[ks][u] = l0[u]; ql[ks][4 + u] = l1[u];
            }
        }
    }

    f32x16 u[4];                    // upd[slot = acc_row(r, h)][d = 32 n + l31]
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) u[n][r] = 0.f;
    float rs[16];                   // per-lane partial row sums: slot acc_row(r, h), this lane's location column
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[r] = 0.f;

    // lane-constant LDS addresses
    const int w_off_k = h * KROW + l31 * 8;                       // staging write, k image (+ 2 it rows)
    const int w_off_v = h * VROW + l31 * 8;
    // plane staging: image (lane >> 4): k hi, v hi, k lo, v lo; lane-dependent row stride
    const int pl_img = lane >> 4;
    const int pl_stride = (pl_img & 1) ? VROW : KROW;
    unsigned char* pl_base = ((pl_img & 1) ? wv + (pl_img >> 1) * VPLANE : wl + (pl_img >> 1) * KPLANE) + (lane & 15) * 16;
    const unsigned char* k_rd = wl + l31 * KROW + h * 16;         // dots B operand: row = location l31
    const int i16 = lane & 15, c16 = ((lane >> 4) & 1) * 16;
    const int tr_row = 8 * h + (i16 >> 2), tr_col = c16 + 4 * (i16 & 3);
    const unsigned char* a_rd = wl + tr_row * AROW + tr_col * 2;                 // attn image [loc][slot]
    const unsigned char* v_rd = wv + tr_row * VROW + tr_col * 2;                 // v image [loc][d]
    constexpr float INV = 1.f / (SA * SA);

    for (int tile = first; tile < last; ++tile) {
        // ---- a. this tile's k and v planes -> LDS, then the next tile's rows take off: their data is first
        //         touched at the top of the next iteration, a whole tile of compute later
        if (PLANES) {
            // lane >> 4 selects the image: 0 k hi, 1 v hi, 2 k lo, 3 v lo; row `it`, 16 B at column 8 (lane & 15)
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                *reinterpret_cast<f32x4*>(pl_base + it * pl_stride) = kt[it];
                *reinterpret_cast<f32x4*>(pl_base + (16 + it) * pl_stride) = vt[it];
            }
            if (tile + 1 < last) load_planes(tile + 1);
        } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                f16x4 hi, lo;
                split4(kt[it], hi, lo);
                *reinterpret_cast<f16x4*>(wl + w_off_k + 2 * it * KROW) = hi;
                *reinterpret_cast<f16x4*>(wl + KPLANE + w_off_k + 2 * it * KROW) = lo;
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                f16x4 hi, lo;
                split4(vt[it], hi, lo);
                *reinterpret_cast<f16x4*>(wv + w_off_v + 2 * it * VROW) = hi;
                *reinterpret_cast<f16x4*>(wv + VPLANE + w_off_v + 2 * it * VROW) = lo;
            }
            if (tile + 1 < last) {
                load_tile(kb, tile + 1, kt);
                load_tile(vb, tile + 1, vt);
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- b. dots[slot, loc] = sum_d q[slot][d] k[loc][d]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f16x8 kh = *reinterpret_cast<const f16x8*>(k_rd + ks * 32);
            const f16x8 kl = *reinterpret_cast<const f16x8*>(k_rd + KPLANE + ks * 32);
            s = mfma16(ql[ks], kh, s);
            s = mfma16(qh[ks], kl, s);
            s = mfma16(qh[ks], kh, s);
        }

        // ---- c. softmax over slots (16 registers x 2 lane halves), + eps
        float x[16];
        float mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = acc_row(r, h) < p.Ks ? s[r] * INV : NEG_BIG;
            mx = fmaxf(mx, x[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = acc_row(r, h) < p.Ks ? expf(x[r] - mx) : 0.f;
            sum += x[r];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float rsm = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = acc_row(r, h) < p.Ks ? x[r] * rsm + p.eps : 0.f;
            rs[r] += x[r];
        }
        if (p.attn_out) {
            float* ao = p.attn_out + (size_t)b * p.Ks * p.N + (size_t)tile * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc_row(r, h) < p.Ks) ao[(size_t)acc_row(r, h) * p.N] = x[r];
        }

        // ---- d. attn planes -> LDS image [location l31][slot], over the k image (its reads are complete: the
        //         wave's LDS operations execute in order): registers 4 g .. 4 g + 3 = slots 8 g + 4 h ..
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f16x4 hi, lo;
            split4(f32x4{x[4 * g4], x[4 * g4 + 1], x[4 * g4 + 2], x[4 * g4 + 3]}, hi, lo);
            unsigned char* dst = wl + l31 * AROW + (8 * g4 + 4 * h) * 2;
            *reinterpret_cast<f16x4*>(dst) = hi;
            *reinterpret_cast<f16x4*>(dst + APLANE) = lo;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- e. upd[slot, d] += sum_loc attn[slot, loc] v[loc, d]   (2 k-steps of 16 locations)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const f16x8 ah = tr_frag(a_rd + kk * 16 * AROW, AROW);
            const f16x8 al = tr_frag(a_rd + APLANE + kk * 16 * AROW, AROW);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f16x8 vh = tr_frag(v_rd + kk * 16 * VROW + n * 64, VROW);
                const f16x8 vl = tr_frag(v_rd + VPLANE + kk * 16 * VROW + n * 64, VROW);
                u[n] = mfma16(al, vh, u[n]);
                u[n] = mfma16(ah, vl, u[n]);
                u[n] = mfma16(ah, vh, u[n]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- workgroup reduction over the 4 waves (fixed order), then one record per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);                  // [wave][slot][d]  4 x 16 KiB
    float* rsum = red + 4 * 32 * SD;                             // [wave][slot]
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            red[(wave * 32 + acc_row(r, h)) * SD + 32 * n + l31] = u[n][r] * INV;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float tot = half_sum32(rs[r]);
        if (l31 == 0) rsum[wave * 32 + acc_row(r, h)] = tot;
    }
    __syncthreads();
    float part[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = t + 256 * i;
        part[i] = (red[e] + red[32 * SD + e]) + (red[2 * 32 * SD + e] + red[3 * 32 * SD + e]);
    }
    float prow = 0.f;
    if (t < 32) prow = (rsum[t] + rsum[32 + t]) + (rsum[64 + t] + rsum[96 + t]);
    __syncthreads();                                             // red / rsum are dead: reuse LDS below
    float* inv = reinterpret_cast<float*>(lds);                  // [32]
    unsigned* flag = reinterpret_cast<unsigned*>(lds) + 32;

    if (p.nrec > 1) {
        float* rec = p.rec + ((size_t)b * p.nrec + g) * REC;
#pragma unroll
        for (int i = 0; i < 16; ++i) rec[t + 256 * i] = part[i];
        if (t < 32) rec[SD * 32 + t] = prow;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
        __syncthreads();
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(p.cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned is_last = ticket == (unsigned)(p.nrec - 1);
            if (is_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                p.cnt[b] = 0;       // self-cleaning: nobody else touches this word any more in this launch, and the
            }                       // next launch finds it zero without a memset node in front of every iteration
            *flag = is_last;
        }
        __syncthreads();
        if (*flag == 0) return;                                   // workgroup-uniform
        // last arriver of this sample: add the records in record order (deterministic), plain loads
        const float* base = p.rec + (size_t)b * p.nrec * REC;
#pragma unroll
        for (int i = 0; i < 16; ++i) part[i] = 0.f;
        prow = 0.f;
        for (int r = 0; r < p.nrec; ++r) {
            const float* rr = base + (size_t)r * REC;
#pragma unroll
            for (int i = 0; i < 16; ++i) part[i] += rr[t + 256 * i];
            if (t < 32) prow += rr[SD * 32 + t];
        }
        __syncthreads();                                          // everyone has read *flag
    }
    if (t < 32) inv[t] = t < p.Ks ? 1.0f / prow : 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = t + 256 * i, slot = e >> 7;
        if (slot < p.Ks) p.updates[(size_t)b * p.Ks * SD + e] = part[i] * inv[slot];
    }
}

// Work split: ONE round of workgroups over the 256 CUs (a workgroup pays ~10 us of prologue + reduction
// around its walk, so long walks and few partial records beat many short ones).  Workgroups per sample =
// 256 / B (at least 1, at most one tile per wave), tiles (32 locations) per wave follow.
inline void pick_split(int B, int N, int& tpw, int& nrec) {
    const int ntiles = N / 32;
    int want = 256 / (B < 1 ? 1 : B);
    if (want < 1) want = 1;
    const int most = (ntiles + 3) / 4;
    if (want > most) want = most;
    tpw = (ntiles + 4 * want - 1) / (4 * want);
    nrec = (ntiles + 4 * tpw - 1) / (4 * tpw);
}

inline size_t cnt_bytes(int B) { return ((size_t)B * 4 + CNT_BYTES - 1) / CNT_BYTES * CNT_BYTES; }

}  // namespace

extern "C" size_t tocvp_slot_attn_ws_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    int tpw, nrec;
    pick_split(B, N, tpw, nrec);
    return cnt_bytes(B) + (size_t)B * nrec * REC * sizeof(float);
}

extern "C" int tocvp_slot_attn_iter_f32(const float* q, const float* k, const float* v, int ldkv,
                                        float* updates, float* attn_out, int B, int Ks, int N,
                                        int D, float scale, float eps, void* ws, size_t ws_bytes,
                                        void* stream) {
    TOCVP_CHECK_ARG(q && k && v && updates && ws);
    TOCVP_CHECK_ARG(B >= 0 && B <= 65535 && Ks > 0 && Ks <= 32 && D == SD);
    TOCVP_CHECK_ARG(N >= 32 && (N % 32) == 0 && ldkv >= D);
    TOCVP_CHECK_ARG((size_t)N * ldkv * 4 < 0x7fffffffu);            // 32-bit byte offsets inside a sample
    if ((ldkv & 3) || !tocvp_aligned16(q) || !tocvp_aligned16(k) || !tocvp_aligned16(v) || !tocvp_aligned16(ws))
        return TOCVP_EALIGN;
    if (B == 0) return TOCVP_OK;
    int tpw, nrec;
    pick_split(B, N, tpw, nrec);
    TOCVP_CHECK_ARG(ws_bytes >= tocvp_slot_attn_ws_bytes(B, N));
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned* cnt = static_cast<unsigned*>(ws);
    float* rec = reinterpret_cast<float*>(static_cast<char*>(ws) + cnt_bytes(B));
    SaArgs p{q, k, v, ldkv, attn_out, rec, cnt, updates, B, Ks, N, tpw, nrec, scale, eps};
    hipLaunchKernelGGL(slot_attn_kernel<false>, dim3(nrec, B), dim3(256), 0, s, p);
    return tocvp_launch_status();
}

extern "C" int tocvp_slot_attn_ws_init(void* ws, size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(ws && ws_bytes > 0);
    return hipMemsetAsync(ws, 0, ws_bytes, static_cast<hipStream_t>(stream)) == hipSuccess ? TOCVP_OK : TOCVP_ELAUNCH;
}

extern "C" int tocvp_slot_attn_iter_planes_f32(const float* q, const void* kv_planes, float* updates,
                                               float* attn_out, int B, int Ks, int N, int D, float scale,
                                               float eps, void* ws, size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(q && kv_planes && updates && ws);
    TOCVP_CHECK_ARG(B >= 0 && B <= 65535 && Ks > 0 && Ks <= 32 && D == SD);
    TOCVP_CHECK_ARG(N >= 32 && (N % 32) == 0);
    TOCVP_CHECK_ARG((size_t)N * 1024 < 0x7fffffffu);
    if (!tocvp_aligned16(q) || !tocvp_aligned16(kv_planes) || !tocvp_aligned16(ws)) return TOCVP_EALIGN;
    if (B == 0) return TOCVP_OK;
    int tpw, nrec;
    pick_split(B, N, tpw, nrec);
    TOCVP_CHECK_ARG(ws_bytes >= tocvp_slot_attn_ws_bytes(B, N));
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned* cnt = static_cast<unsigned*>(ws);
    float* rec = reinterpret_cast<float*>(static_cast<char*>(ws) + cnt_bytes(B));
    // one plane row = 1 KiB = 256 floats: the kernel's row arithmetic is in floats
    SaArgs p{q, static_cast<const float*>(kv_planes), nullptr, 256, attn_out, rec, cnt, updates, B, Ks, N, tpw,
             nrec, scale, eps};
    hipLaunchKernelGGL(slot_attn_kernel<true>, dim3(nrec, B), dim3(256), 0, s, p);
    return tocvp_launch_status();
}
