// Multi-head attention on fp32 MFMA (flash-style online softmax), sequences up to a few hundred
// tokens: predictor self-attention (T <= 300, dh 64), text cross-attention (Tk <= 50),
// SAVi transition (T = K slots, dh 32), text encoder (key-padding lengths, dh 32).
//
// gfx950 mapping
//  * one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.
//  * scores are computed TRANSPOSED, S^T = K Q^T (keys on accumulator rows, queries on lanes):
//    the softmax statistics of a query are then an in-lane reduction over 16 registers plus ONE
//    cross-half shuffle, and the exponentiated tile is directly the B operand (one fp32 per lane)
//    of the next MFMA  O^T += V^T P^T  -- no LDS round trip, no transposition of P.
//  * K/V are streamed in 32-key tiles through LDS (coalesced 16-byte global loads); Q is staged
//    once.  Padded strides (dh+4) make every ds_read_b128 conflict-free.
#include <stdlib.h>

#include "common.h"

namespace {

struct MhaArgs {
    const float* Q; int ldq;
    const float* K; int ldk;
    const float* V; int ldv;
    float* O; int ldo;
    int B, H, Tq, Tk;
    float scale;
    const int32_t* key_len;
    void* Osplit; int nsplit;      // optional (B*Tq, nsplit, H*dh) bf16-plane output instead of O
    const float* bias;             // optional additive score bias (H, Tq, Tk), e.g. T5 relative positions
    int xcd;                       // XCD-aware workgroup ids (TOCVP_MHA_XCD, default 1)
    int TqTot;                     // rows per batch of the Q / O tensors (>= Tq: the launch may cover the first Tq rows only)
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr float NEG_BIG = -1.0e30f;

// Exponentials of the online softmax.
//  * QK16 (the default arithmetic, round 5): ONE v_exp_f32 each, in the log2 domain -- p = 2^(fma(score, k2, -max k2)) with
//    k2 = scale log2(e) on the raw accumulator scores (no per-score multiplication; the argument carries the one rounding of
//    the fma).  Rounds 3-4 used the library expf's operations here too (13, then 9 instructions per exponential: 40 % of the
//    vector instructions of a kernel bound by their issue, profiles/r04_mha.md, r05_mha.md); against a float64 softmax the
//    two forms are equally accurate.  The plane-input kernel (attn_planes.hip) uses the same expressions: the two kernels
//    agree bit for bit.
//  * exact-fp32 products (TOCVP_ATTN_QK=fp32 / TOCVP_PRECISION=fp32, the range-free fallback): the library expf's operations
//    for x <= 0 without its two range selects (same operations in the same order, so the same bits as expf): ph + t =
//    x log2(e) in two floats, 2^(ph + t - e) by v_exp_f32, scaled by 2^e.  One v_max keeps a masked score's difference
//    (-1e30) inside the float -> int conversion; the result is 0 either way.
__device__ __forceinline__ float mha_exp_lib(float x) {
#pragma clang fp contract(off)   // ph must be the ROUNDED product in ph - e, as in the library
    x = __builtin_fmaxf(x, -150.f);
    const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;
    const float ph = x * c;
    float t = __builtin_fmaf(x, c, -ph);
    t = __builtin_fmaf(x, cc, t);
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + t;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}

// QK16: BOTH products run on the f16 matrix cores with split operands (hi + lo fp16 planes of 2^8 x, three
// v_mfma_f32_32x32x16_f16 per 16-deep step, fp32-class; arithmetic of gemm_bf16.hip Elem<true>, |q|, |k|,
// |v| < 255): per 32-key tile and 32 queries at dh = 64, 12 + 12 matrix instructions of 32 cycles instead of
// 32 + 32 of 64 cycles.
//  * scores S^T = K Q^T: the Q / K images hold two 2-byte planes per element (same footprint and row stride
//    as fp32), the accumulator layout does not depend on the operand type, so the softmax code is shared;
//  * O^T += V^T P^T: the exponentiated score tile is used as the B operand straight from its accumulator
//    registers (registers 8s..8s+7 -> fragment of 16-key step s, split into hi / lo in registers); that
//    fixes the key order inside a step, and the A operand (V^T, image transposed while staged) follows it.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_qk(float* row, int c, f32x4 v, int DH, bool qk16) {
    if (!qk16) {
        *reinterpret_cast<f32x4*>(row + c) = v;
        return;
    }
    h16x4 hi, lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float X = __builtin_amdgcn_fmed3f(v[u] * TOCVP_F16X3_ACT_SCALE, -65504.f, 65504.f);
        hi[u] = (_Float16)X;
        lo[u] = (_Float16)(X - (float)hi[u]);
    }
    unsigned char* b = reinterpret_cast<unsigned char*>(row);
    *reinterpret_cast<h16x4*>(b + c * 2) = hi;                 // [DH hi | DH lo | 16 B pad] per row
    *reinterpret_cast<h16x4*>(b + DH * 2 + c * 2) = lo;
}

// NW = waves per workgroup = 32-query blocks per workgroup: 4 (128 queries, default) or 2 (64 queries, round 4,
// opt-in).  At 300 tokens three 128-query workgroups hold 128 + 128 + 44 queries -- the third runs as long as the others
// with two of its four waves idle (22 % of the query slots) -- five 64-query workgroups hold 320 slots, yet measure
// slower (see mha_launch).
template <int DH, bool QK16, int NW>
__global__ __launch_bounds__(64 * NW) void mha_f32_kernel(MhaArgs p) {
    constexpr int QB = 32 * NW, NT = 64 * NW;    // queries / threads per workgroup
    constexpr int QS = DH + 4;          // padded row stride of Q / K tiles (floats; = 2 fp16 planes + pad)
    constexpr int F4 = DH / 4;          // float4 per row
    constexpr int ND = DH / 32;         // 32-wide blocks of the head dim
    constexpr int VTB = 136;            // QK16: bytes per row of the transposed V image [dh][32 keys hi | lo | pad]: 34 words -- the 32 rows a P V
                                        // fragment read touches start in 32 different even banks (36 words: 16 banks, two-way conflicts)
    __shared__ __attribute__((aligned(16))) float lds[QB * QS + 32 * QS + (QK16 ? DH * (VTB / 4) : 32 * DH)];
    float* Qs = lds;
    float* Ks = lds + QB * QS;
    float* Vs = Ks + 32 * QS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // Linear workgroup ids are dealt round-robin over the 8 XCDs (private L2 each): the query blocks of one
    // (batch, head) get ids with the same value mod 8, so the keys / values they all stream are fetched into ONE L2
    // (three 128-query blocks at 300 tokens read the same 154 KB; dealt in launch order they hit three XCDs).
    const int gx = (p.Tq + QB - 1) / QB;
    const int bh = p.xcd ? (blockIdx.x / (8 * gx)) * 8 + (blockIdx.x & 7) : blockIdx.x / gx;
    if (bh >= p.B * p.H) return;
    const int b = bh / p.H, head = bh % p.H;
    const int q0 = (p.xcd ? (blockIdx.x >> 3) % gx : blockIdx.x % gx) * QB;

    const float* Qb = p.Q + (size_t)b * p.TqTot * p.ldq + head * DH;
    const float* Kb = p.K + (size_t)b * p.Tk * p.ldk + head * DH;
    const float* Vb = p.V + (size_t)b * p.Tk * p.ldv + head * DH;
    float* Ob = p.O ? p.O + (size_t)b * p.TqTot * p.ldo + head * DH : nullptr;

    int kv_len = p.Tk;
    if (p.key_len) {
        kv_len = p.key_len[b];
        kv_len = kv_len < 1 ? 1 : (kv_len > p.Tk ? p.Tk : kv_len);
    }

    // ---- stage the query rows (zeros past Tq)
    for (int i = t; i < QB * F4; i += NT) {
        const int r = i / F4, c = (i % F4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + r < p.Tq) v = *reinterpret_cast<const f32x4*>(Qb + (size_t)(q0 + r) * p.ldq + c);
        store_qk(Qs + r * QS, c, v, DH, QK16);
    }

    const bool active = (q0 + wave * 32) < p.Tq;   // wave-uniform
    float m_run = NEG_BIG, l_run = 0.f;
    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

    const int nkb = (kv_len + 31) / 32;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc1 = QK16 ? p.scale * (1.f / (TOCVP_F16X3_ACT_SCALE * TOCVP_F16X3_ACT_SCALE)) : p.scale;
    const float k2 = sc1 * LOG2E;                                  // log2-domain scale of the raw scores
    // K / V tiles are prefetched one tile ahead in registers (clamped, always-valid addresses; rows past Tk
    // are zeroed when stored): the global-memory latency of tile kb+1 runs under the products of tile kb.
    constexpr int NITM = (32 * F4 + NT - 1) / NT;                  // items per thread (4 waves): DH 64 -> 2, DH 32 -> 1
    // PAIR (dh 64, four waves, split operands): the two items of a thread are keys 2j, 2j + 1 at the SAME head-dim quad, so
    // that the transposed V image takes whole 4-byte words [key 2j | key 2j + 1] -- 8 ds_write_b32 per thread and tile
    // instead of 16 ds_write_b16 landing in four banks (PMC before: 61 % of the LDS cycles of this kernel were bank conflicts)
    constexpr bool PAIR = QK16 && NITM == 2 && 32 * F4 == 2 * NT;
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    auto item = [&](int it, int& r, int& c) {                      // -> false: no such item in this thread
        if (PAIR) {
            r = 2 * (t >> 4) + it;
            c = (t & 15) * 4;
            return true;
        }
        const int i = t + NT * it;
        r = min(i, 32 * F4 - 1) / F4;
        c = (min(i, 32 * F4 - 1) % F4) * 4;
        return i < 32 * F4;
    };
    f32x4 kreg[NITM], vreg[NITM];
    auto kv_load = [&](int kb) {
#pragma unroll
        for (int it = 0; it < NITM; ++it) {
            int r, c;
            item(it, r, c);                                        // clamped: always a valid address
            const int key = min(kb * 32 + r, p.Tk - 1);
            kreg[it] = *reinterpret_cast<const f32x4*>(Kb + (size_t)key * p.ldk + c);
            vreg[it] = *reinterpret_cast<const f32x4*>(Vb + (size_t)key * p.ldv + c);
        }
    };
    kv_load(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();   // previous tile fully consumed (also orders the Q staging)
        _Float16 vhi[NITM][4], vlo[NITM][4];
#pragma unroll
        for (int it = 0; it < NITM; ++it) {
            int r, c;
            if (!item(it, r, c)) continue;
            const bool valid = kb * 32 + r < p.Tk;
            f32x4 kv = kreg[it], vv = vreg[it];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kv[u] = valid ? kv[u] : 0.f;
                vv[u] = valid ? vv[u] : 0.f;
            }
            store_qk(Ks + r * QS, c, kv, DH, QK16);
            if (QK16) {
                // transposed fp16 planes of 2^8 v: VT[d][key], so that a lane of the P V product reads
                // runs of 4 consecutive keys of ITS head-dim row
                unsigned char* vt = reinterpret_cast<unsigned char*>(Vs);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float X = __builtin_amdgcn_fmed3f(vv[u] * TOCVP_F16X3_ACT_SCALE, -65504.f, 65504.f);
                    const _Float16 hi = (_Float16)X;
                    vhi[it][u] = hi;
                    vlo[it][u] = (_Float16)(X - (float)hi);
                    if (!PAIR) {
                        *reinterpret_cast<_Float16*>(vt + (c + u) * VTB + r * 2) = vhi[it][u];
                        *reinterpret_cast<_Float16*>(vt + (c + u) * VTB + 64 + r * 2) = vlo[it][u];
                    }
                }
            } else {
                *reinterpret_cast<f32x4*>(Vs + r * DH + c) = vv;
            }
        }
        if (PAIR) {
            unsigned char* vt = reinterpret_cast<unsigned char*>(Vs) + (t & 15) * 4 * VTB + (t >> 4) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                *reinterpret_cast<h16x2*>(vt + u * VTB) = h16x2{vhi[0][u], vhi[NITM - 1][u]};
                *reinterpret_cast<h16x2*>(vt + u * VTB + 64) = h16x2{vlo[0][u], vlo[NITM - 1][u]};
            }
        }
        kv_load(min(kb + 1, nkb - 1));
        __syncthreads();
        if (!active) continue;

        // S^T tile: rows = 32 keys, cols (lanes) = this wave's 32 queries
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if (QK16) {
            const unsigned char* ka = reinterpret_cast<const unsigned char*>(Ks + l31 * QS) + h * 16;
            const unsigned char* qa = reinterpret_cast<const unsigned char*>(Qs + (wave * 32 + l31) * QS) + h * 16;
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) {
                const h16x8 ah = *reinterpret_cast<const h16x8*>(ka + ks * 32);
                const h16x8 al = *reinterpret_cast<const h16x8*>(ka + ks * 32 + DH * 2);
                const h16x8 bh = *reinterpret_cast<const h16x8*>(qa + ks * 32);
                const h16x8 bl = *reinterpret_cast<const h16x8*>(qa + ks * 32 + DH * 2);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, s, 0, 0, 0);
            }
        } else {
            const float* ka = Ks + l31 * QS + 4 * h;
            const float* qa = Qs + (wave * 32 + l31) * QS + 4 * h;
#pragma unroll
            for (int j = 0; j < DH / 8; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ka + 8 * j);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(qa + 8 * j);
#pragma unroll
                for (int u = 0; u < 4; ++u) s = mfma32(a[u], bq[u], s);
            }
        }

        // online softmax over keys: in-lane over 16 regs + the other lane half.  The key mask is applied in the last
        // tile only and the bias branch is taken once per tile, not once per register.  Without a bias the scores stay RAW
        // (maximum, fma(score, k2, -max k2), v_exp_f32); with one (T5's relative positions) they move to the log2 domain first.
        float bm = NEG_BIG, alpha, ps = 0.f;
        const bool mask_tile = kb * 32 + 32 > kv_len;   // wave-uniform: only the last tile holds keys past the end
        if (p.bias || !QK16) {
            // scaled scores (+ bias): log2 domain + v_exp_f32 under QK16, natural domain + the library's operations otherwise
            const float ksc = QK16 ? k2 : sc1, kb_ = QK16 ? LOG2E : 1.f;
            const int q = min(q0 + wave * 32 + l31, p.Tq - 1);
            const float* brow = p.bias ? p.bias + ((size_t)head * p.Tq + q) * p.Tk : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + acc_row(r, h);
                float sv = s[r] * ksc;
                if (brow && key < kv_len) sv += brow[key] * kb_;
                s[r] = sv;
            }
            if (mask_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (kb * 32 + acc_row(r, h) < kv_len) ? s[r] : NEG_BIG;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) bm = fmaxf(bm, s[r]);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float m_new = fmaxf(m_run, bm);
            alpha = QK16 ? __builtin_amdgcn_exp2f(m_run - m_new) : mha_exp_lib(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = QK16 ? __builtin_amdgcn_exp2f(s[r] - m_new) : mha_exp_lib(s[r] - m_new);
                ps += s[r];
            }
            m_run = m_new;
        } else {
            if (mask_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (kb * 32 + acc_row(r, h) < kv_len) ? s[r] : NEG_BIG;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) bm = fmaxf(bm, s[r]);
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float m_new = fmaxf(m_run, bm);
            const float mk = m_new * k2;
            alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, k2, -mk));   // first tile / masked scores: 2^(-1e25) = 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], k2, -mk));
                ps += s[r];
            }
            m_run = m_new;
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        // the running maximum settles after the first tiles: the accumulators (AGPRs: read, multiply, write back = 83
        // instructions) are rescaled only when some query of the wave saw a new maximum (alpha = 1 exactly otherwise)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }

        // O^T (dh x 32 queries) += V^T (dh x keys) * P^T (keys x queries); P^T is `s` as it stands
        if (QK16) {
            // f16x3: registers 8ks .. 8ks+7 of the score tile ARE the B fragment of 16-key step ks, with
            // element j of lane half h = key 16ks + 8(j >> 2) + 4h + (j & 3); the A fragment (V^T) takes
            // the same keys: two runs of 4 consecutive keys of head-dim row l31.
            const unsigned char* vt = reinterpret_cast<const unsigned char*>(Vs);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float P = s[8 * ks + j] * TOCVP_F16X3_ACT_SCALE;      // in [0, 256]
                    ph[j] = (_Float16)P;
                    pl[j] = (_Float16)(P - (float)ph[j]);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const unsigned char* va = vt + (d * 32 + l31) * VTB + (16 * ks + 4 * h) * 2;
                    const h16x4 h0 = *reinterpret_cast<const h16x4*>(va), h1 = *reinterpret_cast<const h16x4*>(va + 16);
                    const h16x4 l0 = *reinterpret_cast<const h16x4*>(va + 64), l1 = *reinterpret_cast<const h16x4*>(va + 80);
                    const h16x8 vh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const h16x8 vl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc[d], 0, 0, 0);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc[d], 0, 0, 0);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc[d], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* va = Vs + acc_row(r, h) * DH + l31;
#pragma unroll
                for (int d = 0; d < ND; ++d) oacc[d] = mfma32(va[32 * d], s[r], oacc[d]);
            }
        }
    }

    // ---- epilogue: normalise, transpose through this wave's own Q rows in LDS, coalesced store
    if (active) {
        const float inv = (QK16 ? 1.f / (TOCVP_F16X3_ACT_SCALE * TOCVP_F16X3_ACT_SCALE) : 1.f) / l_run;
        float* os = Qs + (wave * 32) * QS;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) os[l31 * QS + d * 32 + acc_row(r, h)] = oacc[d][r] * inv;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < 32 * F4; i += 64) {
            const int r = i / F4, c = (i % F4) * 4;
            const int q = q0 + wave * 32 + r;
            if (q >= p.Tq) continue;
            f32x4 o = *reinterpret_cast<const f32x4*>(os + r * QS + c);
            if (p.Osplit == nullptr) {
                *reinterpret_cast<f32x4*>(Ob + (size_t)q * p.ldo + c) = o;
            } else {
                const int E = p.H * DH;
                const int planes = p.nsplit == 22 ? 2 : p.nsplit;
                tocvp_store_planes4(p.Osplit, ((size_t)b * p.TqTot + q) * planes * E + head * DH + c,
                                    (size_t)E, o, p.nsplit);
            }
        }
    }
}

// ONE query row per (sample, head): row `q_row` of every sample against all keys, exact fp32 on the vector ALUs -- the tail of
// sequence lengths that are one past a multiple of the 128-query tile (the ViT's 256 patches + class token,
// timm_encoders.py:59-70: a third query tile for ONE row staged all 257 keys and values of every (sample, head) again).
// A wave per (sample, head): lanes over keys for the scores (each lane reads whole 256-byte key rows), softmax by wave
// reductions, lanes over the head dimension for the output (coalesced value rows, probabilities broadcast from LDS).
constexpr int ONEQ_MAXK = 1024;
template <int DH>
__global__ __launch_bounds__(256) void mha_one_query_kernel(MhaArgs p, int q_row) {
    __shared__ float qs[4][64];
    __shared__ float ps[4][ONEQ_MAXK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= p.B * p.H) return;
    const int b = bh / p.H, head = bh % p.H;
    const float* Kb = p.K + (size_t)b * p.Tk * p.ldk + head * DH;
    const float* Vb = p.V + (size_t)b * p.Tk * p.ldv + head * DH;
    int kv_len = p.Tk;
    if (p.key_len) {
        kv_len = p.key_len[b];
        kv_len = kv_len < 1 ? 1 : (kv_len > p.Tk ? p.Tk : kv_len);
    }
    if (lane < DH) qs[wave][lane] = p.Q[((size_t)b * p.TqTot + q_row) * p.ldq + head * DH + lane] * p.scale;
    __builtin_amdgcn_wave_barrier();
    float mx = NEG_BIG;
    for (int j = lane; j < kv_len; j += 64) {
        const float* kr = Kb + (size_t)j * p.ldk;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int c = 0; c < DH; c += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(kr + c), bq = *reinterpret_cast<const f32x4*>(kr + c + 4);
            const f32x4 qa = *reinterpret_cast<const f32x4*>(&qs[wave][c]), qb = *reinterpret_cast<const f32x4*>(&qs[wave][c + 4]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d0 = fmaf(a[u], qa[u], d0);
                d1 = fmaf(bq[u], qb[u], d1);
            }
        }
        const float sc = d0 + d1;
        ps[wave][j] = sc;
        mx = fmaxf(mx, sc);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < kv_len; j += 64) {
        const float e = __expf(ps[wave][j] - mx);
        ps[wave][j] = e;
        sum += e;
    }
    sum = wave_sum64(sum);
    __builtin_amdgcn_wave_barrier();
    if (lane < DH) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int j = 0;
        for (; j + 4 <= kv_len; j += 4) {
            a0 = fmaf(ps[wave][j], Vb[(size_t)j * p.ldv + lane], a0);
            a1 = fmaf(ps[wave][j + 1], Vb[(size_t)(j + 1) * p.ldv + lane], a1);
            a2 = fmaf(ps[wave][j + 2], Vb[(size_t)(j + 2) * p.ldv + lane], a2);
            a3 = fmaf(ps[wave][j + 3], Vb[(size_t)(j + 3) * p.ldv + lane], a3);
        }
        for (; j < kv_len; ++j) a0 = fmaf(ps[wave][j], Vb[(size_t)j * p.ldv + lane], a0);
        const float o = ((a0 + a1) + (a2 + a3)) / sum;
        if (p.Osplit == nullptr) {
            p.O[((size_t)b * p.TqTot + q_row) * p.ldo + head * DH + lane] = o;
        } else {
            // fp16 operand planes of 2^8 o (the expressions of tocvp_store_planes4, one element per lane)
            const int E = p.H * DH;
            const float X = __builtin_amdgcn_fmed3f(o * TOCVP_F16X3_ACT_SCALE, -65504.f, 65504.f);
            const _Float16 hi = (_Float16)X;
            _Float16* dst = static_cast<_Float16*>(p.Osplit) + ((size_t)b * p.TqTot + q_row) * 2 * E + head * DH + lane;
            dst[0] = hi;
            dst[E] = (_Float16)(X - (float)hi);
        }
    }
}

}  // namespace

static int mha_launch(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                      float* O, int ldo, void* Osplit, int nsplit, int B, int H, int Tq, int Tk,
                      int dh, float scale, const int32_t* key_len, void* stream,
                      const float* bias = nullptr, bool qk16 = false, int tq_tot = 0) {
    TOCVP_CHECK_ARG(Q && K && V && (O || Osplit));
    TOCVP_CHECK_ARG(B >= 0 && H > 0 && Tq > 0 && Tk > 0);
    TOCVP_CHECK_ARG(dh == 32 || dh == 64);
    TOCVP_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && (Osplit || ldo >= H * dh));
    TOCVP_CHECK_ARG(B <= 65535 && H <= 65535);
    TOCVP_CHECK_ARG(Osplit == nullptr || nsplit == 2 || nsplit == 3 || nsplit == 22);
    if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (O && (ldo & 3)) || !tocvp_aligned16(Q) ||
        !tocvp_aligned16(K) || !tocvp_aligned16(V) || (O && !tocvp_aligned16(O)) ||
        (Osplit && !tocvp_aligned16(Osplit)))
        return TOCVP_EALIGN;
    if (B == 0) return TOCVP_OK;
    const int xcd = 1;                             // XCD-aware workgroup ids (the A/B switch TOCVP_MHA_XCD of round 3 is gone)
    TOCVP_CHECK_ARG(tq_tot == 0 || (tq_tot >= Tq && bias == nullptr));
    MhaArgs p{Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Tq, Tk, scale, key_len, Osplit, nsplit, bias, xcd, tq_tot ? tq_tot : Tq};
    // (64-query workgroups -- NW = 2 -- were built in round 4, measured slower (245 vs 194 us at 128 x 8 x 300 x 300) and retired)
    const dim3 grid((unsigned)((size_t)(((long)B * H + 7) / 8) * 8 * ((Tq + 127) / 128)));
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TOCVP_MHA_GO(DH_, QK_) hipLaunchKernelGGL((mha_f32_kernel<DH_, QK_, 4>), grid, dim3(256), 0, s, p)
    if (dh == 64) {
        if (qk16) TOCVP_MHA_GO(64, true);
        else TOCVP_MHA_GO(64, false);
    } else {
        if (qk16) TOCVP_MHA_GO(32, true);
        else TOCVP_MHA_GO(32, false);
    }
#undef TOCVP_MHA_GO
    return tocvp_launch_status();
}

extern "C" int tocvp_mha_f32(const float* Q, int ldq, const float* K, int ldk, const float* V,
                             int ldv, float* O, int ldo, int B, int H, int Tq, int Tk, int dh,
                             float scale, const int32_t* key_len, void* stream) {
    return mha_launch(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, 0, B, H, Tq, Tk, dh, scale, key_len,
                      stream);
}

extern "C" int tocvp_mha_qk16_f32(const float* Q, int ldq, const float* K, int ldk, const float* V,
                                  int ldv, float* O, int ldo, int B, int H, int Tq, int Tk, int dh,
                                  float scale, const int32_t* key_len, void* stream) {
    return mha_launch(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, 0, B, H, Tq, Tk, dh, scale, key_len,
                      stream, nullptr, true);
}

extern "C" int tocvp_mha_qk16_rows_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                                       int ldo, int B, int H, int Tq_total, int q_rows, int Tk, int dh, float scale,
                                       const int32_t* key_len, void* stream) {
    TOCVP_CHECK_ARG(q_rows > 0 && q_rows <= Tq_total);
    return mha_launch(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, 0, B, H, q_rows, Tk, dh, scale, key_len, stream, nullptr, true,
                      Tq_total);
}

static int tocvp_mha_one_query_out(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                   void* Osplit, int B, int H, int Tq_total, int q_row, int Tk, int dh, float scale,
                                   const int32_t* key_len, void* stream);

extern "C" int tocvp_mha_qk16_rows_split_f16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                             void* Osplit, int B, int H, int Tq_total, int q_rows, int Tk, int dh,
                                             float scale, const int32_t* key_len, void* stream) {
    TOCVP_CHECK_ARG(Osplit != nullptr && q_rows > 0 && q_rows <= Tq_total);
    return mha_launch(Q, ldq, K, ldk, V, ldv, nullptr, 0, Osplit, 22, B, H, q_rows, Tk, dh, scale, key_len, stream, nullptr,
                      true, Tq_total);
}

extern "C" int tocvp_mha_one_query_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                                       int ldo, int B, int H, int Tq_total, int q_row, int Tk, int dh, float scale,
                                       const int32_t* key_len, void* stream) {
    return tocvp_mha_one_query_out(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, B, H, Tq_total, q_row, Tk, dh, scale, key_len, stream);
}

extern "C" int tocvp_mha_one_query_split_f16(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                             void* Osplit, int B, int H, int Tq_total, int q_row, int Tk, int dh, float scale,
                                             const int32_t* key_len, void* stream) {
    TOCVP_CHECK_ARG(Osplit != nullptr);
    return tocvp_mha_one_query_out(Q, ldq, K, ldk, V, ldv, nullptr, H * dh, Osplit, B, H, Tq_total, q_row, Tk, dh, scale, key_len,
                                   stream);
}

static int tocvp_mha_one_query_out(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                   void* Osplit, int B, int H, int Tq_total, int q_row, int Tk, int dh, float scale,
                                   const int32_t* key_len, void* stream) {
    TOCVP_CHECK_ARG(Q && K && V && (O || Osplit));
    TOCVP_CHECK_ARG(B >= 0 && H > 0 && Tq_total > 0 && q_row >= 0 && q_row < Tq_total && Tk > 0 && Tk <= ONEQ_MAXK);
    TOCVP_CHECK_ARG(dh == 32 || dh == 64);
    TOCVP_CHECK_ARG(ldq >= H * dh && ldk >= H * dh && ldv >= H * dh && ldo >= H * dh);
    if ((ldk & 3) || !tocvp_aligned16(K)) return TOCVP_EALIGN;
    if (B == 0) return TOCVP_OK;
    MhaArgs p{Q, ldq, K, ldk, V, ldv, O, ldo, B, H, 1, Tk, scale, key_len, Osplit, Osplit ? 22 : 0, nullptr, 0, Tq_total};
    const dim3 grid((unsigned)(((long)B * H + 3) / 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dh == 64) hipLaunchKernelGGL(mha_one_query_kernel<64>, grid, dim3(256), 0, s, p, q_row);
    else hipLaunchKernelGGL(mha_one_query_kernel<32>, grid, dim3(256), 0, s, p, q_row);
    return tocvp_launch_status();
}

extern "C" int tocvp_mha_split_bf16(const float* Q, int ldq, const float* K, int ldk, const float* V,
                                    int ldv, void* Osplit, int nsplit, int B, int H, int Tq, int Tk,
                                    int dh, float scale, const int32_t* key_len, void* stream) {
    TOCVP_CHECK_ARG(Osplit != nullptr);
    return mha_launch(Q, ldq, K, ldk, V, ldv, nullptr, 0, Osplit, nsplit, B, H, Tq, Tk, dh, scale,
                      key_len, stream);
}

extern "C" int tocvp_mha_bias_f32(const float* Q, int ldq, const float* K, int ldk, const float* V,
                                  int ldv, float* O, int ldo, int B, int H, int Tq, int Tk, int dh,
                                  float scale, const int32_t* key_len, const float* bias,
                                  void* stream) {
    TOCVP_CHECK_ARG(bias != nullptr);
    return mha_launch(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, 0, B, H, Tq, Tk, dh, scale, key_len,
                      stream, bias);
}
