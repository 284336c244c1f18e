// Backward of multi-head softmax attention for the predictor training step (SURVEY.md section 8f rank 2;
// reference: torch.autograd through MetaAttention.attention, models/Blocks/attention.py:157-176), fused:
// the score and probability matrices never reach HBM.
//
//   S = scale Q K^T,  P = softmax_keys(S) (keys >= key_len[b] masked),  O = P V          (forward)
//   dV = P^T dO,  dP = dO V^T,  dS = scale P o (dP - delta),  delta_i = <dO_i, O_i>,  dQ = dS K,  dK = dS^T Q
//
// Exact fp32 matrix cores (v_mfma_f32_32x32x2_f32), head dimension 64, 32 x 32 score tiles.  Three launches:
//   stats : per query row  lse_i = log sum_j exp(S_ij)  and  delta_i            (one wave per 32 queries)
//   dkv   : a wave OWNS 32 keys (K_j, V_j as B-operand registers, dK_j / dV_j accumulators) and walks the
//           query tiles, which the 4 waves of the workgroup share through LDS
//   dq    : a wave OWNS 32 queries (Q_i, dO_i as B-operand registers) and walks the key tiles
// so every reduction stays inside one wave: no atomics, no cross-wave sums, deterministic.
//
// The trick that avoids every transpose: the C/D layout of the 32 x 32 MFMA (lane = column, register r of
// lane-half h = row acc_row(r, h)) IS the A-operand layout of a product that reduces over the tile's ROWS
// two at a time (k-slice = rows acc_row(r, 0) and acc_row(r, 1) = register r of the two halves).  A score tile
// held as (rows = queries, columns = keys) therefore feeds dV += P^T dO and dK += dS^T Q straight from its
// accumulator registers, and the transposed tile (rows = keys, columns = queries) feeds dQ += dS K -- which is
// why dkv computes S and dq computes S^T (256 MFMAs per tile pair in total instead of the 160 of an
// implementation that could transpose for free).  With queries on the columns the row statistics of the
// softmax are per-LANE scalars (stats, dq); with queries on the rows they come from a small LDS table (dkv).
#include "common.h"

namespace {

constexpr int DH = 64, TS = 32, LDT = 66;          // head dim, tile side, padded LDS row (floats)
constexpr int TILE = TS * LDT;                      // floats per staged 32 x 64 tile
constexpr float LOG2E = 1.4426950408889634f;

struct AbArgs {
    const float* q; const float* k; const float* v; const float* o; const float* d_o;
    float* dq; float* dk; float* dv; float* stats; const int32_t* key_len;
    int B, H, Tq, Tk, E; float scale;
};

// One 32 x 64 tile (rows row0 .., head columns) of a (T, E) matrix -> registers of the whole workgroup
// (256 threads, 2 float4 each; rows >= T give zeros) and from there into the padded LDS image.
struct TileRegs { f32x4 v[2]; };
__device__ __forceinline__ void tile_fetch(TileRegs& r, const float* base, int row0, int T, int E, int t) {
    const int row = t >> 3, c4 = (t & 7) * 4;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        r.v[u] = row0 + row < T ? *reinterpret_cast<const f32x4*>(base + (size_t)(row0 + row) * E + c4 + 32 * u)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void tile_stash(float* dst, const TileRegs& r, int t) {
    const int row = t >> 3, c4 = (t & 7) * 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {                  // 8-byte stores: the 264-byte row pitch is not 16-byte aligned
        float* d = dst + row * LDT + c4 + 32 * u;
        *reinterpret_cast<float2*>(d) = float2{r.v[u][0], r.v[u][1]};
        *reinterpret_cast<float2*>(d + 2) = float2{r.v[u][2], r.v[u][3]};
    }
}
// A-operand fetch "lane = tile row": X[l31][2 s + h]  (conflict-free on the 66-float pitch)
__device__ __forceinline__ float tile_row_elem(const float* tile, int l31, int h, int s) {
    return tile[l31 * LDT + 2 * s + h];
}
// B-operand fetch "lane = head column": X[acc_row(r, h)][l31 + 32 half]
__device__ __forceinline__ float tile_col_elem(const float* tile, int l31, int h, int r, int half) {
    return tile[acc_row(r, h) * LDT + l31 + 32 * half];
}
// B-operand registers of every wave's OWN tile: b[s] = X[row0(wave) + l31][2 s + h].  The workgroup's four
// tiles (128 consecutive rows) are fetched with coalesced 16-byte loads into four LDS tile slots and read back
// in operand order (a lane reading every other float of its own row from global costs 32 x 32 cache lines).
__device__ __forceinline__ void own_rows(float (&b)[32], float* slots, const float* base, int row0_wg, int T, int E,
                                         int t, int wave, int l31, int h) {
    TileRegs r[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) tile_fetch(r[w], base, row0_wg + w * TS, T, E, t);
#pragma unroll
    for (int w = 0; w < 4; ++w) tile_stash(slots + w * TILE, r[w], t);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 32; ++s) b[s] = tile_row_elem(slots + wave * TILE, l31, h, s);
    __syncthreads();                               // the slots are reused by the caller
}
__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// stats: grid (B * H, ceil(Tq / 128)); wave w of a workgroup owns queries i0 = (blockIdx.y * 4 + w) * 32 ..
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_stats_kernel(AbArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int kl = p.key_len ? min(p.key_len[b], p.Tk) : p.Tk;
    const float* Q = p.q + (size_t)b * p.Tq * p.E + hd * DH;
    const float* Kp = p.k + (size_t)b * p.Tk * p.E + hd * DH;
    const int i0 = (blockIdx.y * 4 + wave) * TS;
    const bool active = i0 < p.Tq;                                  // wave-uniform
    float qb[32];
    own_rows(qb, lds, Q, blockIdx.y * 4 * TS, p.Tq, p.E, t, wave, l31, h);
    const float sc2 = p.scale * LOG2E;
    float m_run = -INFINITY, l_run = 0.f;
    const int nj = (kl + TS - 1) / TS;
    TileRegs kr;
    tile_fetch(kr, Kp, 0, p.Tk, p.E, t);
    for (int j = 0; j < nj; ++j) {
        float* Ks = lds + (j & 1) * TILE;
        tile_stash(Ks, kr, t);
        __syncthreads();
        if (j + 1 < nj) tile_fetch(kr, Kp, (j + 1) * TS, p.Tk, p.E, t);
        if (!active) continue;
        f32x16 st;
        zero(st);
#pragma unroll
        for (int s = 0; s < 32; ++s) st = mfma32(tile_row_elem(Ks, l31, h, s), qb[s], st);    // S^T: rows keys, cols queries
        float m_new = m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = j * TS + acc_row(r, h) < kl ? st[r] * sc2 : -INFINITY;
            m_new = fmaxf(m_new, st[r]);
        }
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += m_new == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(st[r] - m_new);
        l_run = (m_run == -INFINITY ? 0.f : l_run * __builtin_amdgcn_exp2f(m_run - m_new)) + add;
        m_run = m_new;
    }
    if (!active) return;
    // the two lane halves saw disjoint key rows of every tile: merge them
    const float m_o = __shfl_xor(m_run, 32, 64), l_o = __shfl_xor(l_run, 32, 64);
    const float m = fmaxf(m_run, m_o);
    const float l = (m_run == -INFINITY ? 0.f : l_run * __builtin_amdgcn_exp2f(m_run - m)) + (m_o == -INFINITY ? 0.f : l_o * __builtin_amdgcn_exp2f(m_o - m));
    // delta_i = <dO_i, O_i>: half h sums head columns 32 h .. 32 h + 31 of the lane's query row
    float d = 0.f;
    if (i0 + l31 < p.Tq) {
        const size_t off = ((size_t)b * p.Tq + i0 + l31) * p.E + hd * DH + 32 * h;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.d_o + off + c);
            const f32x4 o = *reinterpret_cast<const f32x4*>(p.o + off + c);
            d += a[0] * o[0] + a[1] * o[1] + a[2] * o[2] + a[3] * o[3];
        }
    }
    d += __shfl_xor(d, 32, 64);
    if (h == 0 && i0 + l31 < p.Tq) {
        float* st = p.stats + ((size_t)blockIdx.x * p.Tq + i0 + l31) * 2;
        st[0] = m + log2f(l);                    // log2-domain log-sum-exp of scale * S
        st[1] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// dkv: grid (B * H, ceil(Tk / 128)); wave w owns keys j0 = (blockIdx.y * 4 + w) * 32 ..
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AbArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (2 * TILE + 2 * TS)];
    static_assert(2 * (2 * TILE + 2 * TS) >= 4 * TILE, "four tile slots for own_rows");
    constexpr int STAGE = 2 * TILE + 2 * TS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int kl = p.key_len ? min(p.key_len[b], p.Tk) : p.Tk;
    const float* Q = p.q + (size_t)b * p.Tq * p.E + hd * DH;
    const float* DO = p.d_o + (size_t)b * p.Tq * p.E + hd * DH;
    const float* Kp = p.k + (size_t)b * p.Tk * p.E + hd * DH;
    const float* Vp = p.v + (size_t)b * p.Tk * p.E + hd * DH;
    const float* ST = p.stats + (size_t)blockIdx.x * p.Tq * 2;
    const int j0 = (blockIdx.y * 4 + wave) * TS;
    const bool active = j0 < kl;                                    // wave-uniform; masked keys get zero gradients
    float kb[32], vb[32];
    own_rows(kb, lds, Kp, blockIdx.y * 4 * TS, p.Tk, p.E, t, wave, l31, h);
    own_rows(vb, lds, Vp, blockIdx.y * 4 * TS, p.Tk, p.E, t, wave, l31, h);
    f32x16 dk[2], dv[2];
    zero(dk[0]); zero(dk[1]); zero(dv[0]); zero(dv[1]);
    const float sc2 = p.scale * LOG2E;
    const bool key_ok = j0 + l31 < kl;
    const int ni = (p.Tq + TS - 1) / TS;
    TileRegs qr, gr;
    float sr = 0.f;
    auto fetch = [&](int i) {
        tile_fetch(qr, Q, i * TS, p.Tq, p.E, t);
        tile_fetch(gr, DO, i * TS, p.Tq, p.E, t);
        if (t < 2 * TS) sr = i * TS + (t >> 1) < p.Tq ? ST[(size_t)(i * TS) * 2 + t] : 0.f;
    };
    fetch(0);
    for (int i = 0; i < ni; ++i) {
        float* Qs = lds + (i & 1) * STAGE;
        float* Gs = Qs + TILE;
        float* Ss = Gs + TILE;
        tile_stash(Qs, qr, t);
        tile_stash(Gs, gr, t);
        if (t < 2 * TS) Ss[t] = sr;
        __syncthreads();
        if (i + 1 < ni) fetch(i + 1);
        if (!active) continue;
        f32x16 s_acc, p_acc;
        zero(s_acc); zero(p_acc);
#pragma unroll
        for (int s = 0; s < 32; ++s) s_acc = mfma32(tile_row_elem(Qs, l31, h, s), kb[s], s_acc);   // rows queries, cols keys
#pragma unroll
        for (int s = 0; s < 32; ++s) p_acc = mfma32(tile_row_elem(Gs, l31, h, s), vb[s], p_acc);   // dP = dO V^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, h);
            const bool ok = key_ok && i * TS + row < p.Tq;
            const float pr = ok ? __builtin_amdgcn_exp2f(s_acc[r] * sc2 - Ss[2 * row]) : 0.f;
            s_acc[r] = pr;                                                        // P
            p_acc[r] = pr * (p_acc[r] - Ss[2 * row + 1]) * p.scale;               // dS
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                dv[half] = mfma32(s_acc[r], tile_col_elem(Gs, l31, h, r, half), dv[half]);   // dV += P^T dO
                dk[half] = mfma32(p_acc[r], tile_col_elem(Qs, l31, h, r, half), dk[half]);   // dK += dS^T Q
            }
    }
    if (j0 >= p.Tk) return;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = j0 + acc_row(r, h);
            if (row < p.Tk) {
                const size_t off = ((size_t)b * p.Tk + row) * p.E + hd * DH + 32 * half + l31;
                p.dk[off] = dk[half][r];
                p.dv[off] = dv[half][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// dq: grid (B * H, ceil(Tq / 128)); wave w owns queries i0 = (blockIdx.y * 4 + w) * 32 ..
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AbArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int kl = p.key_len ? min(p.key_len[b], p.Tk) : p.Tk;
    const float* Q = p.q + (size_t)b * p.Tq * p.E + hd * DH;
    const float* DO = p.d_o + (size_t)b * p.Tq * p.E + hd * DH;
    const float* Kp = p.k + (size_t)b * p.Tk * p.E + hd * DH;
    const float* Vp = p.v + (size_t)b * p.Tk * p.E + hd * DH;
    const int i0 = (blockIdx.y * 4 + wave) * TS;
    const bool active = i0 < p.Tq;
    float qb[32], gb[32];
    own_rows(qb, lds, Q, blockIdx.y * 4 * TS, p.Tq, p.E, t, wave, l31, h);
    own_rows(gb, lds, DO, blockIdx.y * 4 * TS, p.Tq, p.E, t, wave, l31, h);
    const bool q_ok = i0 + l31 < p.Tq;
    const float* st = p.stats + ((size_t)blockIdx.x * p.Tq + min(i0 + l31, p.Tq - 1)) * 2;
    const float lse = st[0], delta = st[1];
    f32x16 dq[2];
    zero(dq[0]); zero(dq[1]);
    const float sc2 = p.scale * LOG2E;
    const int nj = (kl + TS - 1) / TS;
    TileRegs kr, vr;
    tile_fetch(kr, Kp, 0, p.Tk, p.E, t);
    tile_fetch(vr, Vp, 0, p.Tk, p.E, t);
    for (int j = 0; j < nj; ++j) {
        float* Ks = lds + (j & 1) * 2 * TILE;
        float* Vs = Ks + TILE;
        tile_stash(Ks, kr, t);
        tile_stash(Vs, vr, t);
        __syncthreads();
        if (j + 1 < nj) {
            tile_fetch(kr, Kp, (j + 1) * TS, p.Tk, p.E, t);
            tile_fetch(vr, Vp, (j + 1) * TS, p.Tk, p.E, t);
        }
        if (!active) continue;
        f32x16 st_acc, pt_acc;
        zero(st_acc); zero(pt_acc);
#pragma unroll
        for (int s = 0; s < 32; ++s) st_acc = mfma32(tile_row_elem(Ks, l31, h, s), qb[s], st_acc);  // rows keys, cols queries
#pragma unroll
        for (int s = 0; s < 32; ++s) pt_acc = mfma32(tile_row_elem(Vs, l31, h, s), gb[s], pt_acc);  // dP^T = V dO^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = q_ok && j * TS + acc_row(r, h) < kl;
            const float pr = ok ? __builtin_amdgcn_exp2f(st_acc[r] * sc2 - lse) : 0.f;
            st_acc[r] = pr * (pt_acc[r] - delta) * p.scale;                       // dS^T
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int half = 0; half < 2; ++half)
                dq[half] = mfma32(st_acc[r], tile_col_elem(Ks, l31, h, r, half), dq[half]);          // dQ += dS K
    }
    if (!active) return;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + acc_row(r, h);
            if (row < p.Tq) p.dq[((size_t)b * p.Tq + row) * p.E + hd * DH + 32 * half + l31] = dq[half][r];
        }
}

}  // namespace

extern "C" int tocvp_attn_bwd_f32(const float* q, const float* k, const float* v, const float* o,
                                  const float* d_o, float* dq, float* dk, float* dv, float* stats,
                                  const int32_t* key_len, int B, int H, int Tq, int Tk, int E, float scale,
                                  void* stream) {
    TOCVP_CHECK_ARG(q && k && v && o && d_o && dq && dk && dv && stats);
    TOCVP_CHECK_ARG(B >= 0 && H > 0 && Tq > 0 && Tk > 0 && E == H * DH);
    TOCVP_CHECK_ARG((long)B * H <= 0x7fffffffL);
    if (!tocvp_aligned16(q) || !tocvp_aligned16(k) || !tocvp_aligned16(v) || !tocvp_aligned16(o) ||
        !tocvp_aligned16(d_o))
        return TOCVP_EALIGN;
    if (B == 0) return TOCVP_OK;
    AbArgs a{q, k, v, o, d_o, dq, dk, dv, stats, key_len, B, H, Tq, Tk, E, scale};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 gq(B * H, (Tq + 127) / 128), gk(B * H, (Tk + 127) / 128);
    hipLaunchKernelGGL(attn_bwd_stats_kernel, gq, dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, gk, dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(256), 0, s, a);
    return tocvp_launch_status();
}
