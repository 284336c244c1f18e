// NOT PART OF THE LIBRARY.  Round-5 experiment, kept as a record: a fused predictor MLP for the authors' evaluation batches
// (32-token tiles x slices of the hidden dimension).  CORRECT (bit-identical to the two-GEMM path when unsliced, 1-5e-6 of float64
// otherwise) and 2.2-2.8x SLOWER than the two GEMMs it would replace (one MI355X, HIP events, us):
//     rows      two GEMMs   this kernel
//      600        33.7         67.6
//     2400        66.7 (57.0 with plane hand-over)   143.6
//     9600       172.1        487.7
// Why: every token tile streams ALL 8 MB of weight planes whatever its height; 32-row tiles read 4x the bytes of the 128-row
// kernel (csrc/mlp_fused.hip) -- 600 MB at 2400 rows -- and at these row counts the weights come from the Infinity Cache, not
// from the 4 MB L2 of an XCD: the pair is bound by weight re-streaming, which fusing the two products does not reduce.
// The two GEMMs (64-row tiles: 300 MB at 2400 rows) stay the small-batch path.
// Fused predictor MLP for SMALL and MID-SIZE row counts (round 5): Y = relu(X W1^T + b1) W2^T + b2 + R in ONE launch for the
// nn.Linear -> ReLU -> nn.Linear pairs of the predictor blocks (reference models/Blocks/attention.py:355-359 applied at :395 /
// :521-523, and :428-432 applied at :461-463) when the batch is the authors' (8-32 sequences: 600 .. 11 000 token rows,
// scripts/05_evaluate_TextOCVP_CATER.sh:3-11).  csrc/mlp_fused.hip serves the many-row regime with 128-token tiles that each
// stream all 8 MB of weight planes; below ~11 000 rows its tiles do not fill the chip and the two GEMMs it replaces were the
// faster path -- 42 + 31 us per pair at 2400 rows, 55 % of a rollout's time at 8 sequences.
//
// gfx950 mapping
//  * a workgroup = 4 waves = 32 token rows x ONE SLICE of the hidden dimension (Hd / S hidden units, in chunks of 128):
//    tiles x S workgroups fill the chip from ~600 rows up (2400 rows: 75 tiles x 4 slices);
//  * the X tile sits in LDS ONCE as fp16 operand planes (32 rows x [hi 512 | lo 512], split while staged or copied if the
//    producer wrote planes), XOR-swizzled 16-byte chunks (64 KB); per chunk  H^T = W1[c] X^T  (operands swapped: tokens
//    on the accumulator's lanes) -> bias + ReLU + planes into a 16 KB LDS image -> Y^T += W2[:, c] H^T; every weight fragment
//    comes from L2 in MFMA-fragment order (1 KiB per wave-instruction, straight into the A operand) and is fetched by exactly
//    one wave of the workgroup; 64 + 16 accumulator registers per lane, 80 KB of LDS: two workgroups per CU;
//  * S > 1: the slices of a tile park their raw Y^T accumulators in a per-stream workspace (write-through 16-byte stores),
//    count themselves in, and the LAST arriver adds the records in slice order (deterministic) and runs the epilogue
//    (scale, b2, residual, 16-byte stores straight from the accumulators) -- the protocol of gemm_bf16.hip's split-K.
// Arithmetic: f16x3 (split fp16 operands, three products per product), the hidden activation re-split exactly as the
// up-projection's plane-writing epilogue would; sums over the hidden dimension are associated per slice, so the result
// differs from the two-GEMM path by fp32 re-association (tests/test_kernels_gpu.py::test_mlp_small_*).
#include <stdlib.h>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ME = 512;                  // model width (rows of X, Y)
constexpr int BM = 32;                   // token rows per workgroup
constexpr int HC = 128;                  // hidden units per chunk
constexpr int KS1 = ME / 16;             // 16-deep k-steps of the first product
constexpr int XROW = 2 * ME * 2;         // bytes per token row of the X image [hi | lo]
constexpr int HROW = 2 * HC * 2;         // bytes per token row of the h image
constexpr int XS_BYTES = BM * XROW, HS_BYTES = BM * HROW;
constexpr int REC = BM * ME;             // floats per parked accumulator record (64 KB)
constexpr int WS_CTR_BYTES = 16384;      // 4096 tile counters
constexpr int WS_RECORDS = 1024;         // 64 MB of records
constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;

struct SmallArgs {
    const void* X; int x_split;          // fp32 rows (M, 512) or fp16 planes (M, 2, 512)
    const unsigned char* W1f; const float* b1;   // W1 (Hd, 512) in fragment order, bias (Hd)
    const unsigned char* W2f; const float* b2;   // W2 (512, Hd) in fragment order, bias (512)
    const float* R; int ldr;
    float* Y; int ldy;
    int M, Hd, S;
    float* ws_part; unsigned* ws_ctr;
};

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool XSPLIT, bool HASR>
__global__ __launch_bounds__(256, 2) void mlp_f16x3_small_kernel(SmallArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[XS_BYTES + HS_BYTES];
    unsigned char* const xs = lds;
    unsigned char* const hs = lds + XS_BYTES;
    typedef const __attribute__((address_space(1))) f16x8* gv8;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int S = p.S;
    const int tile = (int)blockIdx.x / S, slice = (int)blockIdx.x % S;
    const int m0 = tile * BM;
    const int nchunk = p.Hd / HC;
    const int c0 = nchunk * slice / S, c1 = nchunk * (slice + 1) / S;       // never empty: S <= nchunk
    const int KS2 = p.Hd / 16;

    // ---- X tile -> LDS planes (once).  Physical 16-byte chunk = logical ^ (row & 15), logical = plane * 64 + k / 8
    if (XSPLIT) {
        const unsigned char* xp = static_cast<const unsigned char*>(p.X);
#pragma unroll
        for (int i = 0; i < (BM * 128) / 256; ++i) {
            const int piece = t + 256 * i, row = piece >> 7, lc = piece & 127;
            const int grow = min(m0 + row, p.M - 1);                        // rows past M repeat the last row (never stored)
            const u32x4 v = *reinterpret_cast<const u32x4*>(xp + ((size_t)grow * 2 * ME + (size_t)lc * 8) * 2);
            *reinterpret_cast<u32x4*>(xs + row * XROW + ((lc ^ (row & 15)) << 4)) = v;
        }
    } else {
        const float* xf = static_cast<const float*>(p.X);
#pragma unroll
        for (int i = 0; i < (BM * ME / 4) / 256; ++i) {
            const int q = t + 256 * i, row = q >> 7, k4 = (q & 127) * 4;    // 4 consecutive k of one row
            const int grow = min(m0 + row, p.M - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(xf + (size_t)grow * ME + k4);
            f16x4 hi, lo;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float X = __builtin_amdgcn_fmed3f(v[u] * SA, -65504.f, 65504.f);
                hi[u] = (_Float16)X;
                lo[u] = (_Float16)(X - (float)hi[u]);
            }
            const int lc = k4 >> 3, sub = (k4 & 4) * 2;                     // logical chunk of the hi plane, byte inside it
            *reinterpret_cast<f16x4*>(xs + row * XROW + ((lc ^ (row & 15)) << 4) + sub) = hi;
            *reinterpret_cast<f16x4*>(xs + row * XROW + (((64 + lc) ^ (row & 15)) << 4) + sub) = lo;
        }
    }
    __syncthreads();

    f32x16 yacc[4];                                                          // Y^T rows (outputs) 128 w + 32 j + .., lanes = tokens
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[j][r] = 0.f;

    const unsigned x15 = (unsigned)(l31 & 15);
    const unsigned char* xrow = xs + l31 * XROW;                             // B fragments of the first product: this lane's token
    const unsigned char* hrow = hs + l31 * HROW;
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr float UNSCALE = 1.f / (SA * SW);

    for (int c = c0; c < c1; ++c) {
        // ---- first product: H^T (128 hidden x 32 tokens) = W1[c] X^T; wave w owns hidden rows 32 w .. 32 w + 31 of the chunk
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
        const unsigned char* w1 = p.W1f + ((size_t)(c * 4 + w) * KS1) * 2048 + lane16;   // [ks][plane][lane] 1 KiB pieces
#pragma unroll 8
        for (int ks = 0; ks < KS1; ++ks) {
            const f16x8 ah = *(gv8)(w1 + (size_t)ks * 2048), al = *(gv8)(w1 + (size_t)ks * 2048 + 1024);
            const unsigned lc = (unsigned)(2 * ks + h);
            const f16x8 bh = *reinterpret_cast<const f16x8*>(xrow + ((lc ^ x15) << 4));
            const f16x8 bl = *reinterpret_cast<const f16x8*>(xrow + (((64u + lc) ^ x15) << 4));
            hacc = mfma16(al, bh, hacc);
            hacc = mfma16(ah, bl, hacc);
            hacc = mfma16(ah, bh, hacc);
        }
        // ---- bias + ReLU + fp16 planes of 2^8 h into the h image: registers 4 g .. 4 g + 3 = hidden 32 w + 8 g + 4 h + 0..3
        {
            const float* b1 = p.b1 + c * HC + w * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b1 + 8 * g);
                f16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v = hacc[4 * g + u] * UNSCALE + bq[u];
                    v = v > 0.f ? v : 0.f;
                    const float X = __builtin_amdgcn_fmed3f(v * SA, -65504.f, 65504.f);
                    hi[u] = (_Float16)X;
                    lo[u] = (_Float16)(X - (float)hi[u]);
                }
                const unsigned lc = (unsigned)(w * 4 + g);                   // logical chunk of the hi plane (8 hidden per chunk)
                *reinterpret_cast<f16x4*>(hs + l31 * HROW + ((lc ^ x15) << 4) + 8 * h) = hi;
                *reinterpret_cast<f16x4*>(hs + l31 * HROW + (((16u + lc) ^ x15) << 4) + 8 * h) = lo;
            }
        }
        __syncthreads();
        // ---- second product: Y^T (512 outputs x 32 tokens) += W2[:, c] H^T; wave w owns outputs 128 w .. 128 w + 127
        const unsigned char* w2 = p.W2f + ((size_t)(4 * w) * KS2 + (size_t)c * 8) * 2048 + lane16;
#pragma unroll
        for (int ks = 0; ks < HC / 16; ++ks) {
            const unsigned lc = (unsigned)(2 * ks + h);
            const f16x8 bh = *reinterpret_cast<const f16x8*>(hrow + ((lc ^ x15) << 4));
            const f16x8 bl = *reinterpret_cast<const f16x8*>(hrow + (((16u + lc) ^ x15) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char* wp = w2 + ((size_t)j * KS2 + ks) * 2048;
                const f16x8 ah = *(gv8)(wp), al = *(gv8)(wp + 1024);
                yacc[j] = mfma16(al, bh, yacc[j]);
                yacc[j] = mfma16(ah, bl, yacc[j]);
                yacc[j] = mfma16(ah, bh, yacc[j]);
            }
        }
        __syncthreads();                                                     // the h image is free for the next chunk
    }

    // ---- S > 1: park, count in, the last arriver adds the slices in slice order
    if (S > 1) {
        {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p.ws_part, 0, WS_RECORDS * REC * 4, 0x00020000);
            const unsigned off0 = (unsigned)((((size_t)tile * S + slice) * REC + (size_t)t * 4) * sizeof(float));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v{yacc[j][4 * g], yacc[j][4 * g + 1], yacc[j][4 * g + 2], yacc[j][4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, off0 + (j * 4 + g) * 4096, 0, 16);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // every storing wave drains its stores
        __syncthreads();
        unsigned* arrived = reinterpret_cast<unsigned*>(lds);                // the images are dead
        if (t == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.ws_ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(S - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.ws_ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            *arrived = old;
        }
        __syncthreads();
        if (*arrived != (unsigned)(S - 1)) return;
        const float* rec0 = p.ws_part + (size_t)tile * S * REC + (size_t)t * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[j][r] = 0.f;
#pragma unroll 1
        for (int sl = 0; sl < S; ++sl) {
            f32x4 v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = *reinterpret_cast<const f32x4*>(rec0 + (size_t)sl * REC + e * 1024);
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) yacc[e >> 2][4 * (e & 3) + u] += v[e][u];
        }
    }

    // ---- epilogue straight from the accumulators: lane = token m0 + l31, registers 4 g .. 4 g + 3 of yacc[j] = outputs
    // 128 w + 32 j + 8 g + 4 h + 0..3: one 16-byte store each
    const int row = m0 + l31;
    if (row < p.M) {
        float* yrow = p.Y + (size_t)row * p.ldy + w * 128 + 4 * h;
        const float* rrow = HASR ? p.R + (size_t)row * p.ldr + w * 128 + 4 * h : nullptr;
        const float* b2 = p.b2 + w * 128 + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(b2 + 32 * j + 8 * g);
                f32x4 o;
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = yacc[j][4 * g + u] * UNSCALE + bq[u];
                if (HASR) {
                    const f32x4 rq = *reinterpret_cast<const f32x4*>(rrow + 32 * j + 8 * g);
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] += rq[u];
                }
                *reinterpret_cast<f32x4*>(yrow + 32 * j + 8 * g) = o;
            }
    }
}

int small_cus() {
    static const int n = []() {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    return n;
}

}  // namespace

extern "C" size_t tocvp_mlp_f16x3_small_ws_bytes(void) { return (size_t)WS_CTR_BYTES + (size_t)WS_RECORDS * REC * sizeof(float); }

extern "C" int tocvp_mlp_f16x3_small_f32(const void* X, int x_split, const void* w1_frag, const float* b1, const void* w2_frag,
                                         const float* b2, const float* R, int ldr, float* Y, int ldy, int M, int E_, int Hd,
                                         void* ws, size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(X && w1_frag && b1 && w2_frag && b2 && Y);
    TOCVP_CHECK_ARG(M >= 0 && E_ == ME && Hd >= HC && (Hd % HC) == 0 && ldy >= ME && (ldy & 3) == 0);
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= ME && (ldr & 3) == 0));
    TOCVP_CHECK_ARG((size_t)M * ME * 4 < 0xffffffffull);
    TOCVP_CHECK_ARG(ws == nullptr || (ws_bytes >= tocvp_mlp_f16x3_small_ws_bytes() && tocvp_aligned16(ws)));
    if (!tocvp_aligned16(X) || !tocvp_aligned16(w1_frag) || !tocvp_aligned16(w2_frag) || !tocvp_aligned16(b1) ||
        !tocvp_aligned16(b2) || !tocvp_aligned16(Y) || (R && !tocvp_aligned16(R)))
        return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    const int tiles = (M + BM - 1) / BM, nchunk = Hd / HC;
    // slices of the hidden dimension until every CU has a workgroup (two fit), at most one chunk per slice
    int S = 1;
    if (ws) {
        static const int smax = []() { const char* e = getenv("TOCVP_MLP_SMALL_SMAX"); return e ? atoi(e) : 16; }();
        while (2 * S <= smax && 2 * S <= nchunk && tiles * S < small_cus() && tiles * 2 * S <= WS_RECORDS &&
               tiles <= WS_CTR_BYTES / 4)
            S *= 2;
    }
    SmallArgs p{X, x_split, static_cast<const unsigned char*>(w1_frag), b1, static_cast<const unsigned char*>(w2_frag), b2, R, ldr,
                Y, ldy, M, Hd, S, ws ? reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + WS_CTR_BYTES) : nullptr,
                static_cast<unsigned*>(ws)};
    const dim3 grid((unsigned)(tiles * S));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (x_split) {
        if (R) hipLaunchKernelGGL((mlp_f16x3_small_kernel<true, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mlp_f16x3_small_kernel<true, false>), grid, dim3(256), 0, s, p);
    } else {
        if (R) hipLaunchKernelGGL((mlp_f16x3_small_kernel<false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mlp_f16x3_small_kernel<false, false>), grid, dim3(256), 0, s, p);
    }
    return tocvp_launch_status();
}
