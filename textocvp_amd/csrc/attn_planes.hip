// Self-attention whose q / k / v arrive as fp16 OPERAND PLANES (round 5): MetaAttention.attention + head split / merge of the
// reference (models/Blocks/attention.py:183-215, 245-265) for the predictor blocks and the ViT blocks
// (EncodersDecoders/timm_encoders.py:59-70), head dim 64.
//
// Why: mha_f32_kernel<64, true, 4> (attn.hip) issues vector instructions for 71 % of its cycles -- ~650 per 32-key tile and
// wave against 24 MFMAs (profiles/r04_mha.md): it splits every K / V element into fp16 hi / lo planes and transposes V while
// it stages the tile, and each of the three 128-query workgroups of a (sample, head) repeats that.  Here the qkv
// projection's epilogue writes the planes ONCE (tocvp_store_planes4: the same hi = f16(2^8 x), lo = f16(2^8 x - hi) the
// attention kernel computed for itself), as rows [hi of all columns | lo of all columns] per token, and this kernel only
// copies them:
//  * K and V tiles of 32 keys go global -> registers -> LDS as 16-byte pieces (no conversion, no transpose), a tile ahead in
//    registers and a tile ahead in a second LDS stage: ONE barrier per tile; the images are lane-linear rows of 256
//    bytes whose conflict-free chunk order comes from XOR swizzles of the source chunk;
//  * the V^T fragments of O^T += V^T P^T come out of the ROW-major V image through ds_read_b64_tr_b16 (hardware
//    transpose);
//  * the Q fragments of S^T = K Q^T stay in registers for the whole kernel (16-byte loads straight from the planes);
// The arithmetic is that of the fp32-input kernel expression for expression -- same split values, same order of the
// matrix products, same online softmax over 32-key tiles -- so the result is BIT-IDENTICAL to tocvp_mha_qk16_f32 on the
// same q / k / v (tests/test_kernels_gpu.py::test_mha_planes_equals_the_fp32_input_kernel): the range-checked pass, which
// keeps fp32 hand-overs and the fp32-input kernel, reproduces the fast path's bits.
// TOCVP_HIPCC_FLAGS: -mllvm -amdgpu-mfma-vgpr-form
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct MhaPArgs {
    const _Float16* Q; int ldq;    // planes (rows, 2, ld*) of 2^8 x: plane stride ld*, row stride 2 ld* (elements); the pointers
    const _Float16* K; int ldk;    // are already offset to the column block of the q / k / v projection
    const _Float16* V; int ldv;
    float* O; int ldo;             // fp32 output rows, or
    void* Osplit;                  // fp16 operand planes (B * TqTot, 2, H * 64) for the output projection
    int B, H, Tq, Tk;
    float scale;
    const int32_t* key_len;
    int TqTot;                     // rows per sample of the Q / O tensors (the launch covers the first Tq of them)
};

constexpr float NEG_BIG = -1.0e30f;

// -DTOCVP_MHAP_STAMP (scripts/probes/mha_planes_stamp.hip): s_memtime at the phase boundaries of wave 0, summed per workgroup
#ifdef TOCVP_MHAP_STAMP
__device__ unsigned long long tocvp_mhap_stamps[16384 * 8];
#define MHAP_T(i)                                                    \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        stamp_acc[i] += now_ - stamp_last;                           \
        stamp_last = now_;                                           \
        __builtin_amdgcn_sched_barrier(0);                           \
    } while (0)
#else
#define MHAP_T(i) do {} while (0)
#endif

// 8 consecutive-k fp16 of this lane's column out of a ROW-major [k][column] image: two hardware-transposed reads of
// 4 k-rows x 16 columns per 16-lane group (cdna_hip_programming.md T10); ``addr`` = this lane's address for rows 0..3,
// the second read takes the rows ``gap`` bytes further down.
__device__ __forceinline__ h16x8 tr_frag(const unsigned char* addr, int gap) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    union { s16x4 s[2]; h16x8 f; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(addr));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(addr + gap));
    return u.f;
}

// LDS images of a 32-key tile: lane-linear rows of 256 bytes [hi 64 | lo 64] = 16 chunks of 16 bytes (1 KiB = 4 key rows
// per wave-instruction of the staging stores, also the shape an LDS-DMA fill would have); the conflict-free order comes
// from the SOURCE side:
//   K: physical chunk c' of key row r holds logical chunk c' ^ (r & 15)  -> the 16 rows of a ds_read_b128 lane group
//      ({0-3, 12-15, 20-27}, ...: 16 distinct r & 15) take 16 distinct slots;
//   V: physical chunk c' holds logical chunk c' ^ ((r & 3) << 2)         -> the four key rows of a transposed read
//      (ds_read_b64_tr_b16: rows q = 0..3, 64 contiguous bytes each) take the four 64-byte quarters of the bank row.
constexpr int KV_IMG = 32 * 256;
constexpr int STAGE = 2 * KV_IMG;
constexpr int OS = 64 + 4;              // floats per row of the epilogue's transposition image

// NW waves = NW 32-query blocks per workgroup.  Exponentials: ONE v_exp_f32 of fma(raw score, k2, -max k2), k2 = scale x
// log2(e) -- the expressions of attn.hip, so the two kernels agree bit for bit.
template <int NW>
__global__ __launch_bounds__(64 * NW) void mha_planes_kernel(MhaPArgs p) {
    constexpr int DH = 64, QB = 32 * NW;
    constexpr int LDS_BYTES = 2 * STAGE > NW * 32 * OS * 4 ? 2 * STAGE : NW * 32 * OS * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // linear workgroup ids are dealt round-robin over the 8 XCDs: the query blocks of one (sample, head) take ids with the
    // same value mod 8, so the keys / values they all stream sit in ONE L2
    const int gx = (p.Tq + QB - 1) / QB;
    const int bh = (blockIdx.x / (8 * gx)) * 8 + (blockIdx.x & 7);
    if (bh >= p.B * p.H) return;
    const int b = bh / p.H, head = bh % p.H;
    const int q0 = ((blockIdx.x >> 3) % gx) * QB;

    int kv_len = p.Tk;
    if (p.key_len) {
        kv_len = p.key_len[b];
        kv_len = kv_len < 1 ? 1 : (kv_len > p.Tk ? p.Tk : kv_len);
    }
    const int nkb = (kv_len + 31) / 32;

    // ---- staging through registers: a 32-key tile = 512 K pieces + 512 V pieces of 16 bytes, thread t moves pieces t and
    // t + 256 of each (key rows t >> 4 and 16 + (t >> 4), physical chunk t & 15); the loads of tile kb + 2 are issued right
    // behind the stores of tile kb + 1 and fly under the products of tile kb.  (LDS-DMA was built and measured: its four
    // wave-instructions per tile cost a wave ~580 cycles to issue and ~520 more waiting for the landing, a third of a tile;
    // profiles/r05_mha.md.)
    const unsigned char* Kb = reinterpret_cast<const unsigned char*>(p.K + (size_t)b * p.Tk * 2 * p.ldk + head * DH);
    const unsigned char* Vb = reinterpret_cast<const unsigned char*>(p.V + (size_t)b * p.Tk * 2 * p.ldv + head * DH);
    const int pr = t >> 4, pcp = t & 15;
    u32x4 kreg[2], vreg[2];
    // byte offsets inside this sample's planes stay below 2^32 (Tk x 4 ld bytes) and the factors below 2^24: one
    // v_mad_u32_u24 per piece on top of the lane-constant chunk offsets (64-bit address arithmetic per piece cost a dozen
    // quarter-rate integer multiplies per tile)
    const unsigned rowb_k = 4u * (unsigned)p.ldk, rowb_v = 4u * (unsigned)p.ldv;       // a key row = 2 planes of 2-byte elements
    unsigned ck_off[2], cv_off[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = pr + 16 * it;
        const int ck = pcp ^ (r & 15), cv = pcp ^ ((r & 3) << 2);              // logical chunks of this physical chunk
        ck_off[it] = (unsigned)((ck >> 3) * p.ldk + (ck & 7) * 8) * 2u;
        cv_off[it] = (unsigned)((cv >> 3) * p.ldv + (cv & 7) * 8) * 2u;
    }
    auto kv_load = [&](int kb) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const unsigned key = (unsigned)min(kb * 32 + pr + 16 * it, p.Tk - 1);   // rows past the end repeat the last key
            kreg[it] = *reinterpret_cast<const u32x4*>(Kb + (size_t)(__umul24(key, rowb_k) + ck_off[it]));
            vreg[it] = *reinterpret_cast<const u32x4*>(Vb + (size_t)(__umul24(key, rowb_v) + cv_off[it]));
        }
    };
    auto kv_write = [&](int stage) {
        unsigned char* ks = lds + stage * STAGE + t * 16;     // piece t = row t >> 4, physical chunk t & 15: lane-linear
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            *reinterpret_cast<u32x4*>(ks + it * 4096) = kreg[it];
            *reinterpret_cast<u32x4*>(ks + KV_IMG + it * 4096) = vreg[it];
        }
    };
    kv_load(0);

    // ---- this lane's query row: the eight B fragments of S^T = K Q^T (4 k-steps x hi / lo), straight from the planes
    const bool active = (q0 + wave * 32) < p.Tq;              // wave-uniform
    h16x8 qh[4], ql[4];
    {
        const int q = min(q0 + wave * 32 + l31, p.Tq - 1);
        const _Float16* qrow = p.Q + ((size_t)b * p.TqTot + q) * 2 * p.ldq + head * DH + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qh[ks] = *reinterpret_cast<const h16x8*>(qrow + ks * 16);
            ql[ks] = *reinterpret_cast<const h16x8*>(qrow + p.ldq + ks * 16);
        }
    }

    kv_write(0);
    kv_load(min(1, nkb - 1));

    float m_run = NEG_BIG, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    const float sc = p.scale * (1.f / (TOCVP_F16X3_ACT_SCALE * TOCVP_F16X3_ACT_SCALE));
    const float k2 = sc * 1.4426950408889634f;                    // log2-domain scale of the raw scores
    // lane-constant parts of the LDS addresses.  K (ds_read_b128, key row l31, logical chunk C0 + h with C0 = plane * 8 + 2 ks
    // even): physical chunk = C0 ^ z, z = h ^ (l31 & 15).  V (transposed reads: lane 4 q + pp of a 16-lane group supplies key
    // row q, columns 4 pp .. 4 pp + 3; the group's lanes receive 16 consecutive columns): 64-byte quarter (2 plane + d) ^ q.
    const int kz = h ^ (l31 & 15);
    const int ka_row = l31 * 256;
    const int vq = (lane & 15) >> 2;
    const int va_row = (4 * h + vq) * 256 + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);

#ifdef TOCVP_MHAP_STAMP
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_readcyclecounter();
    const unsigned long long stamp_first = stamp_last;
#endif
    auto tile = [&](int kb, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const unsigned char* st = lds + (kb & 1) * STAGE;
        // S^T tile: rows = 32 keys, columns (lanes) = this wave's 32 queries
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        h16x8 ah[2], al[2];
        auto kfrag = [&](int ks) {
            ah[ks & 1] = *reinterpret_cast<const h16x8*>(st + ka_row + (((2 * ks) ^ kz) << 4));
            al[ks & 1] = *reinterpret_cast<const h16x8*>(st + ka_row + (((8 + 2 * ks) ^ kz) << 4));
        };
        kfrag(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) kfrag(ks + 1);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks & 1], qh[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks & 1], ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks & 1], qh[ks], s, 0, 0, 0);
        }
        MHAP_T(3);
        // online softmax over the keys: in-lane over the 16 registers + the other lane half
        if (MASKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (kb * 32 + acc_row(r, h) < kv_len) ? s[r] : NEG_BIG;
        }
        float bm = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) bm = fmaxf(bm, s[r]);
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float m_new = fmaxf(m_run, bm);
        const float mk = m_new * k2;
        const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, k2, -mk));   // first tile / masked scores: 2^(-1e25) = 0
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], k2, -mk));
            ps += s[r];
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // the accumulators are rescaled only when some query of the wave saw a new maximum (alpha = 1 exactly otherwise)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        // O^T (dh x 32 queries) += V^T (dh x keys) P^T (keys x queries): registers 8 ks .. 8 ks + 7 of the score tile ARE the B
        // fragment of 16-key step ks, element j of lane half h = key 16 ks + 8 (j >> 2) + 4 h + (j & 3); the transposed
        // reads deliver the A fragment (V^T) in that key order: rows 4 h .. 4 h + 3 and the same eight rows further down
        MHAP_T(4);
        h16x8 vh[2], vl[2];
        auto vfrag = [&](int i) {                                             // i = 2 ks + d
            const unsigned char* va = st + KV_IMG + va_row + (i >> 1) * 16 * 256;
            vh[i & 1] = tr_frag(va + (((i & 1) ^ vq) << 6), 8 * 256);
            vl[i & 1] = tr_frag(va + (((2 + (i & 1)) ^ vq) << 6), 8 * 256);
        };
        vfrag(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { h16x8 v; unsigned u[4]; } ph_, pl_;
#pragma unroll
            for (int j = 0; j < 8; j += 2)                                        // P in [0, 256]: no clamp
                tocvp_split2_f16(s[8 * ks + j] * TOCVP_F16X3_ACT_SCALE, s[8 * ks + j + 1] * TOCVP_F16X3_ACT_SCALE,
                                 ph_.u[j >> 1], pl_.u[j >> 1]);
            const h16x8 ph = ph_.v, pl = pl_.v;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (2 * ks + d < 3) vfrag(2 * ks + d + 1);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[d], ph, oacc[d], 0, 0, 0);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[d], pl, oacc[d], 0, 0, 0);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[d], ph, oacc[d], 0, 0, 0);
            }
        }
        MHAP_T(5);
    };

#ifdef TOCVP_MHAP_STAMP
    stamp_last = __builtin_readcyclecounter();
#endif
    for (int kb = 0; kb < nkb; ++kb) {
        // tile kb is in stage kb & 1 (stored one iteration ago); behind the barrier every wave is done with tile kb - 1,
        // whose stage takes tile kb + 1 from the registers, and the loads of tile kb + 2 start
        MHAP_T(0);
        __syncthreads();
        MHAP_T(1);
        if (kb + 1 < nkb) {
            kv_write((kb + 1) & 1);
            kv_load(min(kb + 2, nkb - 1));
        }
        MHAP_T(2);
        if (!active) continue;
        if (kb * 32 + 32 > kv_len) tile(kb, std::true_type{});               // only the last tile holds keys past the end
        else tile(kb, std::false_type{});
    }

    // ---- epilogue: normalise, transpose through LDS (the stages are free after the barrier), whole rows out
    __syncthreads();
#ifdef TOCVP_MHAP_STAMP
    if (t == 0 && blockIdx.x < 16384) {
        for (int i = 0; i < 6; ++i) tocvp_mhap_stamps[blockIdx.x * 8 + i] = stamp_acc[i];
        tocvp_mhap_stamps[blockIdx.x * 8 + 6] = stamp_first;
        tocvp_mhap_stamps[blockIdx.x * 8 + 7] = __builtin_readcyclecounter();
    }
#endif
    if (!active) return;
    const float inv = (1.f / (TOCVP_F16X3_ACT_SCALE * TOCVP_F16X3_ACT_SCALE)) / l_run;
    float* os = reinterpret_cast<float*>(lds) + wave * 32 * OS;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) os[l31 * OS + d * 32 + acc_row(r, h)] = oacc[d][r] * inv;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 32 * (DH / 4); i += 64) {
        const int r = i / (DH / 4), c = (i % (DH / 4)) * 4;
        const int q = q0 + wave * 32 + r;
        if (q >= p.Tq) continue;
        const f32x4 o = *reinterpret_cast<const f32x4*>(os + r * OS + c);
        if (p.Osplit == nullptr) {
            *reinterpret_cast<f32x4*>(p.O + ((size_t)b * p.TqTot + q) * p.ldo + head * DH + c) = o;
        } else {
            const int E = p.H * DH;
            tocvp_store_planes4(p.Osplit, ((size_t)b * p.TqTot + q) * 2 * E + head * DH + c, (size_t)E, o, 22);
        }
    }
}

}  // namespace

static int planes_args_ok(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, float* O, int ldo,
                          void* Osplit, int B, int H, int Tq_total, int Tk, int dh) {
    TOCVP_CHECK_ARG(Q && K && V && ((O != nullptr) != (Osplit != nullptr)));
    TOCVP_CHECK_ARG(B >= 0 && H > 0 && Tq_total > 0 && Tk > 0 && dh == 64);
    TOCVP_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && (Osplit || ldo >= H * dh));
    TOCVP_CHECK_ARG((long)B * H <= 0x7fffff);
    // 24-bit factors / 32-bit byte offsets inside one sample's planes (attn_planes.hip::kv_load)
    TOCVP_CHECK_ARG(Tk < (1 << 24) && ldk < (1 << 22) && ldv < (1 << 22) && (size_t)Tk * 4 * (ldk > ldv ? ldk : ldv) < 0xffff0000ull);
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (O && (ldo & 3)) || !tocvp_aligned16(Q) || !tocvp_aligned16(K) ||
        !tocvp_aligned16(V) || (O && !tocvp_aligned16(O)) || (Osplit && !tocvp_aligned16(Osplit)))
        return TOCVP_EALIGN;
    return TOCVP_OK;
}

extern "C" int tocvp_mha_planes_f16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, float* O, int ldo,
                                    void* Osplit, int B, int H, int Tq_total, int q_rows, int Tk, int dh, float scale,
                                    const int32_t* key_len, void* stream) {
    const int st = planes_args_ok(Q, ldq, K, ldk, V, ldv, O, ldo, Osplit, B, H, Tq_total, Tk, dh);
    if (st != TOCVP_OK) return st;
    TOCVP_CHECK_ARG(q_rows > 0 && q_rows <= Tq_total);
    if (B == 0) return TOCVP_OK;
    MhaPArgs p{static_cast<const _Float16*>(Q), ldq, static_cast<const _Float16*>(K), ldk, static_cast<const _Float16*>(V), ldv,
               O, ldo, Osplit, B, H, q_rows, Tk, scale, key_len, Tq_total};
    // 4 waves = 128 queries per workgroup (3- and 5-wave workgroups leave fewer empty wave slots at 257 / 300 rows but need
    // three waves per SIMD at 168 registers, which spills inside the tile loop: 490 vs 349 us at 256 x 8 x 300, profiles/r05_mha.md)
    const dim3 grid((unsigned)((size_t)(((long)B * H + 7) / 8) * 8 * ((q_rows + 127) / 128)));
    hipLaunchKernelGGL(mha_planes_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return tocvp_launch_status();
}
