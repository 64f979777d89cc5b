// K2: upfirdn2d (zero-insert upsample -> pad/crop -> 2-D FIR, true convolution -> decimate) for gfx950.
//
// Arithmetic contract (reference upfirdn2d_kernel.cu:85-129, upfirdn2d.py:101-102), per axis:
//   mid = o*down + up - 1 - pad0 ; i0 = floor(mid / up) ; j0 = (i0+1)*up - mid - 1
//   out[o] = sum_{t>=0, j0+t*up < k}  in[i0 + t] * kflip[j0 + t*up],  kflip[j] = k[k-1-j], in[] = 0 outside.
//
// Two implementations:
//   * fir_tile_kernel<UP,DOWN,KH,KW>: minor == 1; a 16x64 output tile per 256-thread block, input tile
//     (with halo) staged once in LDS, flipped taps in registers, 4 consecutive outputs per lane, optional
//     fused bias + leaky-ReLU epilogue.  Instantiated for the configurations the model uses.
//   * fir_direct_kernel: any (up, down, kernel, minor); one output per thread straight from global/L2.
#include "te_common.h"
#include <hip/hip_fp16.h>

namespace {

__host__ __device__ __forceinline__ int fdiv(int a, int b) {  // floor division, b > 0
    int q = a / b;
    return (q * b > a) ? q - 1 : q;
}

struct FirParams {
    int in_h, in_w, out_h, out_w;
    int pad_x0, pad_y0;
    int kh, kw, up_x, up_y, down_x, down_y, minor;
    int64_t major;
    int size_b, act;
    float alpha, scale;
    int ext_x, ext_y;        // blur44: the last tile column / row also covers one extra output column / row
    int tiles_x, tiles_y, zgroups;   // blur44 / fir_tile: 1-D grid, see xcd_tile()
};

// XCD-aware block -> (tile, plane group) mapping of the FIR kernels.  Consecutive workgroup ids go round-robin over the 8
// XCDs (each with its own L2), so with a plain 3-D grid the x / y neighbours of a tile - which share its halo rows and, for
// rows of 2H+1 floats, the partially written cache lines at its left and right edge - always sit behind a different L2 and
// every shared line crosses the fabric twice (measured round 2: 1.38x - 1.66x the algorithmic traffic).  Here the j-th
// block of XCD x takes tile j % T of plane group x + 8 * (j / T): all tiles of a plane run on ONE XCD at about the same
// time and share those lines through its L2; a block then walks the planes pg, pg + zgroups, ... as before.
#ifndef TE_FIR_XCD          // build knob for A/B measurements (tools/exp_build.py): 0 = tile-major ids, as the 3-D grid had them
#define TE_FIR_XCD 1
#endif
struct TileId { int bx, by, pg; };
__device__ __forceinline__ TileId xcd_tile(const FirParams& p) {
    const int T = p.tiles_x * p.tiles_y;
#if TE_FIR_XCD
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t = j % T;
    return TileId{t % p.tiles_x, t / p.tiles_x, xcd + 8 * (j / T)};
#else
    const int t = blockIdx.x % T;
    return TileId{t % p.tiles_x, t / p.tiles_x, (int)(blockIdx.x / T)};
#endif
}
inline unsigned xcd_grid(FirParams& p, int tiles_x, int tiles_y, int64_t z) {
    p.tiles_x = tiles_x; p.tiles_y = tiles_y;
    p.zgroups = (int)(te::cdiv(z, 8) * 8);
    return (unsigned)((int64_t)p.zgroups * tiles_x * tiles_y);
}

template <bool AG = false>
__device__ __forceinline__ float epilogue(float v, const float* b, int ch, const FirParams& p) {
    if constexpr (AG) return v;          // alpha / scale belong to the prologue in this mode
    if (b) v += b[ch];
    if (p.act == 3) v = v > 0.f ? v : v * p.alpha;
    return v * p.scale;
}

constexpr int TOH = 16, TOW = 64;

// AG ("activation gradient" prologue, used by the backward of the fused blur + bias + leaky-ReLU): the staged value is
// x * (ref > 0 ? scale : alpha * scale) with ref the saved forward output, and every block also emits the sum of the
// staged elements it OWNS (first TOH x TOW of its input tile; the last tile of a row / column owns its halo too), i.e.
// the bias-gradient partial of its plane, to partial[plane][tile] (no atomics; the caller adds them up).
template <int UP, int DOWN, int KH, int KW, bool AG = false>
__global__ __launch_bounds__(256) void fir_tile_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                       const float* __restrict__ k, const float* __restrict__ b,
                                                       const FirParams p, const float* __restrict__ ref = nullptr,
                                                       float* __restrict__ partial = nullptr) {
    constexpr int TIH = ((TOH - 1) * DOWN + KH - 1) / UP + 1;
    constexpr int TIW = ((TOW - 1) * DOWN + KW - 1) / UP + 1;
    constexpr int TIWP = (TIW + 3) & ~3;  // row stride, multiple of 4 floats
    constexpr int NTY = (KH + UP - 1) / UP, NTX = (KW + UP - 1) / UP;
    static_assert(KH % UP == 0 && KW % UP == 0, "taps must split evenly over the upsampling phases");
    __shared__ __attribute__((aligned(16))) float sx[TIH * TIWP + 4];
    __shared__ float sk[KH * KW];
    __shared__ float sred[4];

    // flipped taps: registers for UP == 1 (compile-time indices), LDS when the phase selects them at run time
    float kf[KH][KW];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int c = 0; c < KW; ++c) kf[a][c] = k[(KH - 1 - a) * KW + (KW - 1 - c)];
    if (threadIdx.x < KH * KW) sk[threadIdx.x] = k[KH * KW - 1 - threadIdx.x];

    const TileId tid3 = xcd_tile(p);
    const int ox0 = tid3.bx * TOW, oy0 = tid3.by * TOH;
    const int mid_x0 = ox0 * DOWN + UP - 1 - p.pad_x0, mid_y0 = oy0 * DOWN + UP - 1 - p.pad_y0;
    const int ix0 = fdiv(mid_x0, UP), iy0 = fdiv(mid_y0, UP);
    const int ty = threadIdx.x >> 4, tx = (threadIdx.x & 15) * 4;

    // A block walks several planes (plane-group stride) with a register prefetch: the tile of the next plane is loaded
    // while the current one is filtered out of LDS, so the global-load latency is paid once per block, not per plane.
    constexpr int NLD = (TIH * TIW + 255) / 256;
    float stage[NLD];
    float rv[AG ? NLD : 1];
    // all loads of a tile are issued back to back (clamped address + select: no branch, so the NLD global loads of a
    // lane are in flight together instead of one latency after the other)
    auto fetch = [&](int64_t mj) {
        const float* xin = x + (size_t)mj * p.in_h * p.in_w;
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int e = threadIdx.x + 256 * r;
            const int ry = e / TIW, rx = e - ry * TIW;
            const int gy = iy0 + ry, gx = ix0 + rx;
            const bool ok = e < TIH * TIW && gy >= 0 && gy < p.in_h && gx >= 0 && gx < p.in_w;
            const float v = xin[ok ? (size_t)gy * p.in_w + gx : 0];
            stage[r] = ok ? v : 0.f;
        }
        if constexpr (AG) {
            const float* rin = ref + (size_t)mj * p.in_h * p.in_w;
#pragma unroll
            for (int r = 0; r < NLD; ++r) {
                const int e = threadIdx.x + 256 * r;
                const int ry = e / TIW, rx = e - ry * TIW;
                const int gy = iy0 + ry, gx = ix0 + rx;
                const bool ok = e < TIH * TIW && gy >= 0 && gy < p.in_h && gx >= 0 && gx < p.in_w;
                rv[r] = rin[ok ? (size_t)gy * p.in_w + gx : 0];
            }
        }
    };
    if ((int64_t)tid3.pg < p.major) fetch(tid3.pg);

    for (int64_t mj = tid3.pg; mj < p.major; mj += p.zgroups) {
        __syncthreads();                 // the previous plane's filter pass is done with sx / sred
        if constexpr (AG) {
            float own = 0.f;
            const bool last_y = tid3.by == p.tiles_y - 1, last_x = tid3.bx == p.tiles_x - 1;
#pragma unroll
            for (int r = 0; r < NLD; ++r) {
                const int e = threadIdx.x + 256 * r;
                const int ry = e / TIW, rx = e - ry * TIW;
                stage[r] *= rv[r] > 0.f ? p.scale : p.alpha * p.scale;
                if ((ry < TOH || last_y) && (rx < TOW || last_x)) own += stage[r];     // stage is 0 outside the image
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) own += __shfl_down(own, o, 64);
            if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = own;
        }
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int e = threadIdx.x + 256 * r;
            const int ry = e / TIW, rx = e - ry * TIW;
            if (e < TIH * TIW) sx[ry * TIWP + rx] = stage[r];
        }
        __syncthreads();
        if (mj + p.zgroups < p.major) fetch(mj + p.zgroups);       // in flight during the filter pass below
        if constexpr (AG) {
            if (threadIdx.x == 0)
                partial[(size_t)mj * (p.tiles_x * p.tiles_y) + tid3.by * p.tiles_x + tid3.bx] =
                    (sred[0] + sred[1]) + (sred[2] + sred[3]);
        }
        const int oy = oy0 + ty;
        if (oy < p.out_h) {
            float res[4];
            if constexpr (UP == 1 && DOWN == 1 && KH == 4 && KW == 4) {
                // blur fast path: the 4 outputs of this lane need the 4 x 7 window sx[ty..ty+3][tx..tx+6]; read it as
                // two aligned 16-byte LDS loads per row (8 ds_read_b128 instead of 64 conflicting ds_read_b32)
#pragma unroll
                for (int q = 0; q < 4; ++q) res[q] = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float4 w0 = *reinterpret_cast<const float4*>(&sx[(ty + a) * TIWP + tx]);
                    const float4 w1 = *reinterpret_cast<const float4*>(&sx[(ty + a) * TIWP + tx + 4]);
                    const float win[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) res[q] += win[q + c] * kf[a][c];
                }
            } else {
            const int mid_y = mid_y0 + ty * DOWN;
            const int iy = fdiv(mid_y, UP);
            const int jy = (iy + 1) * UP - mid_y - 1;
            const int ry = iy - iy0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mid_x = mid_x0 + (tx + q) * DOWN;
                const int ix = fdiv(mid_x, UP);
                const int jx = (ix + 1) * UP - mid_x - 1;
                const int rx = ix - ix0;
                float acc = 0.f;
#pragma unroll
                for (int a = 0; a < NTY; ++a)
#pragma unroll
                    for (int c = 0; c < NTX; ++c) {
                        float w;
                        if constexpr (UP == 1) w = kf[a][c];
                        else w = sk[(jy + a * UP) * KW + jx + c * UP];   // phase-selected tap (LDS broadcast)
                        acc += sx[(ry + a) * TIWP + rx + c] * w;
                    }
                res[q] = acc;
            }
            }
            float* orow = out + ((size_t)mj * p.out_h + oy) * p.out_w;
            const int ch = b ? (int)(mj % p.size_b) : 0;
            if (ox0 + tx + 3 < p.out_w) {      // one 16-byte store per lane; rows of odd width (257, 129, ...) are only
                                               // 4-byte aligned, which global dwordx4 stores accept
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                f32x4u v;
                v[0] = epilogue<AG>(res[0], b, ch, p); v[1] = epilogue<AG>(res[1], b, ch, p);
                v[2] = epilogue<AG>(res[2], b, ch, p); v[3] = epilogue<AG>(res[3], b, ch, p);
                *reinterpret_cast<f32x4u*>(orow + ox0 + tx) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ox = ox0 + tx + q;
                    if (ox < p.out_w) orow[ox] = epilogue<AG>(res[q], b, ch, p);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void fir_direct_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                         const float* __restrict__ k, const float* __restrict__ b,
                                                         const FirParams p) {
    const int64_t total = p.major * p.out_h * p.out_w * p.minor;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t r = e;
        const int mn = (int)(r % p.minor); r /= p.minor;
        const int ox = (int)(r % p.out_w); r /= p.out_w;
        const int oy = (int)(r % p.out_h);
        const int64_t mj = r / p.out_h;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0, mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int ix0 = fdiv(mid_x, p.up_x), iy0 = fdiv(mid_y, p.up_y);
        const int jx0 = (ix0 + 1) * p.up_x - mid_x - 1, jy0 = (iy0 + 1) * p.up_y - mid_y - 1;
        float acc = 0.f;
        for (int jy = jy0, iy = iy0; jy < p.kh; jy += p.up_y, ++iy) {
            if (iy < 0 || iy >= p.in_h) continue;
            for (int jx = jx0, ix = ix0; jx < p.kw; jx += p.up_x, ++ix) {
                if (ix < 0 || ix >= p.in_w) continue;
                acc += x[(((size_t)mj * p.in_h + iy) * p.in_w + ix) * p.minor + mn] *
                       k[(p.kh - 1 - jy) * p.kw + (p.kw - 1 - jx)];
            }
        }
        out[e] = epilogue(acc, b, b ? (int)(mj % p.size_b) : 0, p);
    }
}

// blocks along z: enough blocks to fill the chip (~24 per CU), the remaining planes are walked inside the block
inline int64_t fir_planes_z(int64_t major, int64_t tiles) {
    return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(major, 32768), te::cdiv(6144, tiles)));
}

template <int UP, int DOWN, int KH, int KW>
void launch_tile(float* out, const float* x, const float* k, const float* b, const FirParams& p, hipStream_t s) {
    const int64_t tiles = te::cdiv(p.out_w, TOW) * te::cdiv(p.out_h, TOH);
    FirParams q = p;
    dim3 grid(xcd_grid(q, (int)te::cdiv(p.out_w, TOW), (int)te::cdiv(p.out_h, TOH), fir_planes_z(p.major, tiles)), 1u, 1u);
    fir_tile_kernel<UP, DOWN, KH, KW><<<grid, 256, 0, s>>>(out, x, k, b, q);
}


// ------------------------------------------------------------------------------------------------------------------
// Blur fast path: up = down = 1, 4 x 4 taps (every Blur of the generator and discriminator, model_spatial_query.py:137-153,
// and their adjoints), optionally with the activation-gradient prologue (AG, see fir_tile_kernel).  HBM-bound: what it
// has to get right is the memory pipeline —
//   * the input tile is fetched as 16-byte loads (buffer_load_dwordx4 at 4-byte alignment: rows of 257 / 129 / ... floats
//     are not 16-byte aligned, the hardware does not care) from a tile origin rounded DOWN to a multiple of 4 columns, so
//     a group of four is either entirely left of the image or starts inside it; only the group that straddles the right
//     edge falls back to guarded 4-byte loads.  4x fewer vector-memory instructions than one dword per lane: at one dword
//     per lane the texture addresser (64 lanes = 16 cycles per instruction) caps a CU near 16 B/clk, i.e. the chip near
//     8 TB/s * efficiency, which is where the dword version sat (3.5 - 4.8 TB/s);
//   * 32 x 64 output tiles (35 x 72 staged elements): 9 % halo rows instead of 19 %;
//   * LDS rows are 128 floats apart: ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
//     (+32), i.e. 8 lanes of one tile row and 8 lanes of the next; with a row stride that is a multiple of 64 banks the
//     two rows' 16-byte slots interleave exactly and every read is conflict-free (the old stride of 68 floats was 2-way:
//     47 - 59 % of LDS cycles were bank conflicts);
//   * the window of a lane starts D = (tile origin - aligned origin) columns into its aligned 16-byte slots; D is a
//     launch constant, so it is a template parameter and the window is picked with static register indices.
constexpr int BOH = 32, BOW = 64;                 // output tile
constexpr int BIH = BOH + 4, BNQ = 18;            // staged rows (33 output rows: see the +1 rule), 16-byte groups per row (72 columns >= 3 + 68)
constexpr int BST = 128;                          // LDS row stride (floats)

typedef float f32x4v __attribute__((ext_vector_type(4)));

// MODE 0: plain (+ bias / leaky-ReLU epilogue); 1: AG, the activation-gradient PROLOGUE (see fir_tile_kernel); 2: the
// activation-gradient EPILOGUE: out = FIR(x) * (ref > 0 ? scale : alpha * scale) with ref the saved forward output at the
// OUTPUT positions, and partial[plane][tile] = sum of the block's outputs (bias gradient): the backward of
// 'conv + bias + leaky-ReLU -> blur' (the discriminator's ResBlock, model_spatial_query.py:744-768, 780-798) in one pass.
template <int MODE, int D, bool EXT>
__global__ __launch_bounds__(256) void blur44_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ k,
                                                     const float* __restrict__ b, const FirParams p,
                                                     const float* __restrict__ ref = nullptr, float* __restrict__ partial = nullptr) {
    __shared__ __attribute__((aligned(16))) float sx[BIH * BST];
    __shared__ float sred[4];
    float kf[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) kf[a][c] = k[(3 - a) * 4 + (3 - c)];

    const TileId tid3 = xcd_tile(p);
    const int ox0 = tid3.bx * BOW, oy0 = tid3.by * BOH;
    const int ix0 = ox0 - p.pad_x0, iy0 = oy0 - p.pad_y0;      // up = down = 1: mid = o - pad, i0 = mid, taps 0..3
    const int ax0 = ix0 - D;                                    // multiple of 4 (D = ix0 & 3, the same for every tile)
    const int ty = threadIdx.x >> 4, tx = (threadIdx.x & 15) * 4;
    const unsigned plane_b = (unsigned)p.in_h * p.in_w * 4u;

    constexpr int NIT = BIH * BNQ, NLD = (NIT + 255) / 256;
    f32x4v stage[NLD];
    constexpr bool AG = (MODE == 1);
    f32x4v rv[AG ? NLD : 1];
    // per-item geometry (the same for every plane).  Every item is ONE unconditional 16-byte buffer load (no branch between
    // the loads of a tile: they are all in flight together): items outside the image get an out-of-range offset (the
    // hardware returns 0), the group that straddles the right edge is loaded 1-3 columns further left (entirely inside
    // the row) and shifted into place with selects when the tile is committed.
    unsigned goff[NLD];
    int lds_off[NLD], sh[NLD];            // sh: -1 no item, 0..3 columns the loaded group sits left of its slot
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int e = threadIdx.x + 256 * r;
        const int ry = e / BNQ, q = e - ry * BNQ;
        const int gy = iy0 + ry, gx = ax0 + 4 * q;
        lds_off[r] = ry * BST + 4 * q;
        sh[r] = e < NIT ? 0 : -1;
        goff[r] = 0x80000000u;
        if (e < NIT && gy >= 0 && gy < p.in_h && gx >= 0 && gx < p.in_w) {
            const int gxs = min(gx, p.in_w - 4);
            sh[r] = gx - gxs;
            goff[r] = (unsigned)(gy * p.in_w + gxs) * 4u;
        }
    }
    auto load_tile = [&](const float* base, int64_t mj, f32x4v* dst) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)mj * p.in_h * p.in_w), 0, plane_b, 0x00020000);
#pragma unroll
        for (int r = 0; r < NLD; ++r) dst[r] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[r], 0, 0));
    };
    const bool edge_x = ax0 + 4 * BNQ > p.in_w;       // block-uniform: only tiles that reach the right edge hold a shifted group
    auto place = [&](f32x4v L, int s_) {      // v[j] = L[j + s] (0 beyond the group): the right-edge group, shifted into its slot
        if (!edge_x) return L;
        f32x4v v;
        v[0] = s_ == 0 ? L[0] : (s_ == 1 ? L[1] : (s_ == 2 ? L[2] : L[3]));
        v[1] = s_ == 0 ? L[1] : (s_ == 1 ? L[2] : (s_ == 2 ? L[3] : 0.f));
        v[2] = s_ == 0 ? L[2] : (s_ == 1 ? L[3] : 0.f);
        v[3] = s_ == 0 ? L[3] : 0.f;
        return v;
    };
    auto fetch = [&](int64_t mj) {
        load_tile(x, mj, stage);
        if constexpr (AG) load_tile(ref, mj, rv);
    };
    if ((int64_t)tid3.pg < p.major) fetch(tid3.pg);

    for (int64_t mj = tid3.pg; mj < p.major; mj += p.zgroups) {
        __syncthreads();                 // the previous plane's filter pass is done with sx / sred
        if constexpr (AG) {
            float own = 0.f;
            const bool last_y = tid3.by == p.tiles_y - 1, last_x = tid3.bx == p.tiles_x - 1;
#pragma unroll
            for (int r = 0; r < NLD; ++r) {
                const int e = threadIdx.x + 256 * r;
                const int ry = e / BNQ, q = e - ry * BNQ;
                stage[r] = place(stage[r], sh[r]);
                const f32x4v rr = place(rv[r], sh[r]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    stage[r][j] *= rr[j] > 0.f ? p.scale : p.alpha * p.scale;
                    const int rx = 4 * q + j - D;                      // column relative to the tile origin
                    if (sh[r] >= 0 && (ry < BOH || last_y) && rx >= 0 && (rx < BOW || last_x)) own += stage[r][j];   // 0 outside the image
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) own += __shfl_down(own, o, 64);
            if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = own;
        }
#pragma unroll
        for (int r = 0; r < NLD; ++r)
            if (sh[r] >= 0) *reinterpret_cast<f32x4v*>(&sx[lds_off[r]]) = AG ? stage[r] : place(stage[r], sh[r]);
        __syncthreads();
        if (mj + p.zgroups < p.major) fetch(mj + p.zgroups);       // in flight during the filter pass below
        if constexpr (AG) {
            if (threadIdx.x == 0)
                partial[(size_t)mj * (p.tiles_x * p.tiles_y) + tid3.by * p.tiles_x + tid3.bx] =
                    (sred[0] + sred[1]) + (sred[2] + sred[3]);
        }
        // lane (ty, tx) filters output rows oy0 + 2 ty, oy0 + 2 ty + 1 (they share 3 of their 4 input rows), columns tx..tx+3.
        // The +1 rule: sizes of the form 64 n + 1 / 32 n + 1 (every (2H+1)-sized tensor of the upsampling path) do not get a
        // tile column / row of their own for the last pixel; the last tile's edge lanes filter a 5th column / 3rd row.
        // (EXT: compiled in only for launches that need it, the extra multiply-adds cost every lane: the kernel is bound by
        // instruction issue, not by HBM)
        constexpr int NH = EXT ? 3 : 2, NQ = EXT ? 5 : 4;
        const bool xcol = EXT && p.ext_x && tid3.bx == p.tiles_x - 1 && tx == BOW - 4;
        const bool xrow = EXT && p.ext_y && tid3.by == p.tiles_y - 1 && ty == 15;
        // MODE 2: the saved forward output at this lane's output positions, requested before the filter pass
        float rf[MODE == 2 ? NH : 1][MODE == 2 ? NQ : 1];
        if constexpr (MODE == 2) {
            const float* rp = ref + (size_t)mj * p.out_h * p.out_w;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int oy = oy0 + 2 * ty + h;
                const bool rowok = oy < p.out_h && !(h == 2 && !xrow);
                const float* rrow = rp + (size_t)(rowok ? oy : 0) * p.out_w;
                if (rowok && ox0 + tx + 3 < p.out_w) {
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                    const f32x4u v = *reinterpret_cast<const f32x4u*>(rrow + ox0 + tx);
                    rf[h][0] = v[0]; rf[h][1] = v[1]; rf[h][2] = v[2]; rf[h][3] = v[3];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) rf[h][q] = (rowok && ox0 + tx + q < p.out_w) ? rrow[ox0 + tx + q] : 0.f;
                }
                if (EXT) rf[h][NQ - 1] = (rowok && xcol) ? rrow[ox0 + tx + 4] : 0.f;
            }
        }
        float res[NH][NQ];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int q = 0; q < NQ; ++q) res[h][q] = 0.f;
        // The extra column / row are BLOCK-uniform branches (last tile column / row only): three quarters of the blocks of a
        // 257-wide plane run the plain 2 x 4 filter (128 multiply-adds per lane) instead of the 3 x 5 one (240) that round 2
        // compiled into every lane of an EXT launch.
        const bool ecol = EXT && p.ext_x && tid3.bx == p.tiles_x - 1;
        const bool erow = EXT && p.ext_y && tid3.by == p.tiles_y - 1;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const float* row = &sx[(2 * ty + a) * BST + tx];
            const f32x4v w0 = *reinterpret_cast<const f32x4v*>(row);
            const f32x4v w1 = *reinterpret_cast<const f32x4v*>(row + 4);
            const f32x4v w2 = *reinterpret_cast<const f32x4v*>(row + 8);
            const float w[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ka = a - h;                 // tap row of output row h
                if (ka < 0 || ka > 3) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) res[h][q] += w[D + q + c] * kf[ka][c];
                if (EXT && ecol) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) res[h][NQ - 1] += w[D + 4 + c] * kf[ka][c];
                }
            }
        }
        if (EXT && erow) {                            // third output row of the last tile row: input rows 2 .. 5
#pragma unroll
            for (int a = 2; a < 6; ++a) {
                const float* row = &sx[(2 * ty + a) * BST + tx];
                const f32x4v w0 = *reinterpret_cast<const f32x4v*>(row);
                const f32x4v w1 = *reinterpret_cast<const f32x4v*>(row + 4);
                const f32x4v w2 = *reinterpret_cast<const f32x4v*>(row + 8);
                const float w[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) res[NH - 1][q] += w[D + q + c] * kf[a - 2][c];
            }
        }
        const int ch = b ? (int)(mj % p.size_b) : 0;
        float own2 = 0.f;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int q = 0; q < NQ; ++q) res[h][q] *= rf[h][q] > 0.f ? p.scale : p.alpha * p.scale;
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int oy = oy0 + 2 * ty + h;
            if (oy >= p.out_h || (h == 2 && !xrow)) continue;
            float* orow = out + ((size_t)mj * p.out_h + oy) * p.out_w;
            if (ox0 + tx + 3 < p.out_w) {      // one 16-byte store per lane (4-byte aligned rows are fine for global stores)
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                f32x4u v;
                v[0] = epilogue<MODE != 0>(res[h][0], b, ch, p); v[1] = epilogue<MODE != 0>(res[h][1], b, ch, p);
                v[2] = epilogue<MODE != 0>(res[h][2], b, ch, p); v[3] = epilogue<MODE != 0>(res[h][3], b, ch, p);
                *reinterpret_cast<f32x4u*>(orow + ox0 + tx) = v;
                if (MODE == 2) own2 += (v[0] + v[1]) + (v[2] + v[3]);
                if (EXT && xcol) {
                    orow[ox0 + tx + 4] = epilogue<MODE != 0>(res[h][NQ - 1], b, ch, p);
                    if (MODE == 2) own2 += res[h][NQ - 1];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ox = ox0 + tx + q;
                    if (ox < p.out_w) {
                        orow[ox] = epilogue<MODE != 0>(res[h][q], b, ch, p);
                        if (MODE == 2) own2 += res[h][q];
                    }
                }
            }
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) own2 += __shfl_down(own2, o, 64);
            if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = own2;
            __syncthreads();             // (the barrier at the top of the loop keeps sred intact until thread 0 has read it)
            if (threadIdx.x == 0)
                partial[(size_t)mj * (p.tiles_x * p.tiles_y) + tid3.by * p.tiles_x + tid3.bx] =
                    (sred[0] + sred[1]) + (sred[2] + sred[3]);
        }
    }
}

inline int blur44_tiles(int n, int t) { return (n > t && n % t == 1) ? n / t : (int)te::cdiv(n, t); }     // the +1 rule

template <int AG>
void launch_blur44(float* out, const float* x, const float* k, const float* b, const FirParams& p, hipStream_t s, const float* ref,
                   float* partial) {
    FirParams q = p;
    const int tx_n = blur44_tiles(p.out_w, BOW), ty_n = blur44_tiles(p.out_h, BOH);
    q.ext_x = tx_n * BOW < p.out_w;
    q.ext_y = ty_n * BOH < p.out_h;
    const int64_t tiles = (int64_t)tx_n * ty_n;
    dim3 grid(xcd_grid(q, tx_n, ty_n, fir_planes_z(p.major, 2 * tiles)), 1u, 1u);
    const int d = (-p.pad_x0) & 3;
    if (q.ext_x || q.ext_y) {
        switch (d) {
            case 0: blur44_kernel<AG, 0, true><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
            case 1: blur44_kernel<AG, 1, true><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
            case 2: blur44_kernel<AG, 2, true><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
            default: blur44_kernel<AG, 3, true><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
        }
    } else {
        switch (d) {
            case 0: blur44_kernel<AG, 0, false><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
            case 1: blur44_kernel<AG, 1, false><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
            case 2: blur44_kernel<AG, 2, false><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
            default: blur44_kernel<AG, 3, false><<<grid, 256, 0, s>>>(out, x, k, b, q, ref, partial); break;
        }
    }
}

}  // namespace

extern "C" int te_blur_actgrad_tiles(int in_h, int in_w, int kh, int kw, int pad_x0, int pad_x1, int pad_y0, int pad_y1) {
    const int oh = in_h + pad_y0 + pad_y1 - kh + 1, ow = in_w + pad_x0 + pad_x1 - kw + 1;
    if (oh <= 0 || ow <= 0) return TE_ERR_SHAPE;
    if (in_w < 4) return (int)(te::cdiv(ow, TOW) * te::cdiv(oh, TOH));     // narrower than one 16-byte group: generic tile kernel
    return blur44_tiles(ow, BOW) * blur44_tiles(oh, BOH);
}

extern "C" int te_blur_actgrad_f32(float* gx, float* partial, const float* g, const float* ref, const float* k, int64_t major,
                                   int in_h, int in_w, int kh, int kw, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                   float alpha, float scale, te_stream_t stream_) {
    TE_REQUIRE(gx && partial && g && ref && k, TE_ERR_NULL, "te_blur_actgrad_f32: NULL pointer");
    TE_REQUIRE(major >= 0 && in_h > 0 && in_w > 0, TE_ERR_SHAPE, "te_blur_actgrad_f32: bad dims");
    TE_REQUIRE(kh == 4 && kw == 4, TE_ERR_UNSUPPORTED, "te_blur_actgrad_f32: 4x4 taps only");
    TE_REQUIRE(pad_x0 >= 0 && pad_x1 >= 0 && pad_y0 >= 0 && pad_y1 >= 0, TE_ERR_UNSUPPORTED,
               "te_blur_actgrad_f32: pads must be >= 0 (every input element has to be staged by some tile)");
    FirParams p{};
    p.in_h = in_h; p.in_w = in_w;
    p.out_h = in_h + pad_y0 + pad_y1 - kh + 1;
    p.out_w = in_w + pad_x0 + pad_x1 - kw + 1;
    TE_REQUIRE(p.out_h > 0 && p.out_w > 0, TE_ERR_SHAPE, "te_blur_actgrad_f32: empty output");
    p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
    p.up_x = p.up_y = p.down_x = p.down_y = 1; p.minor = 1; p.major = major;
    p.size_b = 1; p.act = 0; p.alpha = alpha; p.scale = scale;
    if (major == 0) return 0;
    TE_REQUIRE(major <= 0x7FFFFFFF / 4, TE_ERR_SHAPE, "te_blur_actgrad_f32: too many planes");
    TE_REQUIRE((int64_t)in_h * in_w * 4 < 0x7FFFFFFF, TE_ERR_UNSUPPORTED, "te_blur_actgrad_f32: plane too large");
    if (in_w >= 4) {
        launch_blur44<1>(gx, g, k, nullptr, p, (hipStream_t)stream_, ref, partial);
    } else {
        const int64_t tiles = te::cdiv(p.out_w, TOW) * te::cdiv(p.out_h, TOH);
        dim3 grid(xcd_grid(p, (int)te::cdiv(p.out_w, TOW), (int)te::cdiv(p.out_h, TOH), fir_planes_z(major, tiles)), 1u, 1u);
        fir_tile_kernel<1, 1, 4, 4, true><<<grid, 256, 0, (hipStream_t)stream_>>>(gx, g, k, nullptr, p, ref, partial);
    }
    return te::launch_status("te_blur_actgrad_f32");
}

extern "C" int te_blur_gradact_f32(float* gx, float* partial, const float* g, const float* ref, const float* k, int64_t major,
                                   int in_h, int in_w, int kh, int kw, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                   float alpha, float scale, te_stream_t stream_) {
    TE_REQUIRE(gx && partial && g && ref && k, TE_ERR_NULL, "te_blur_gradact_f32: NULL pointer");
    TE_REQUIRE(major >= 0 && in_h > 0 && in_w >= 4, TE_ERR_SHAPE, "te_blur_gradact_f32: bad dims (in_w >= 4)");
    TE_REQUIRE(kh == 4 && kw == 4, TE_ERR_UNSUPPORTED, "te_blur_gradact_f32: 4x4 taps only");
    FirParams p{};
    p.in_h = in_h; p.in_w = in_w;
    p.out_h = in_h + pad_y0 + pad_y1 - kh + 1;
    p.out_w = in_w + pad_x0 + pad_x1 - kw + 1;
    TE_REQUIRE(p.out_h > 0 && p.out_w > 0, TE_ERR_SHAPE, "te_blur_gradact_f32: empty output");
    p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
    p.up_x = p.up_y = p.down_x = p.down_y = 1; p.minor = 1; p.major = major;
    p.size_b = 1; p.act = 0; p.alpha = alpha; p.scale = scale;
    if (major == 0) return 0;
    TE_REQUIRE(major <= 0x7FFFFFFF / 4, TE_ERR_SHAPE, "te_blur_gradact_f32: too many planes");
    TE_REQUIRE((int64_t)in_h * in_w * 4 < 0x7FFFFFFF, TE_ERR_UNSUPPORTED, "te_blur_gradact_f32: plane too large");
    launch_blur44<2>(gx, g, k, nullptr, p, (hipStream_t)stream_, ref, partial);
    return te::launch_status("te_blur_gradact_f32");
}

// half / double (the reference dispatches its kernel over AT_DISPATCH_FLOATING_TYPES_AND_HALF, upfirdn2d_kernel.cu:57-58,
// 187-189): the direct form for every (up, down, taps); accumulation in fp32 for half, in double for double.
namespace {
template <typename T, typename A>
__global__ __launch_bounds__(256) void fir_direct_any_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ k,
                                                             const FirParams p) {
    const int64_t total = p.major * p.out_h * p.out_w * p.minor;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t r = e;
        const int mn = (int)(r % p.minor); r /= p.minor;
        const int ox = (int)(r % p.out_w); r /= p.out_w;
        const int oy = (int)(r % p.out_h);
        const int64_t mj = r / p.out_h;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0, mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int ix0 = fdiv(mid_x, p.up_x), iy0 = fdiv(mid_y, p.up_y);
        const int jx0 = (ix0 + 1) * p.up_x - mid_x - 1, jy0 = (iy0 + 1) * p.up_y - mid_y - 1;
        A acc = (A)0;
        for (int jy = jy0, iy = iy0; jy < p.kh; jy += p.up_y, ++iy) {
            if (iy < 0 || iy >= p.in_h) continue;
            for (int jx = jx0, ix = ix0; jx < p.kw; jx += p.up_x, ++ix) {
                if (ix < 0 || ix >= p.in_w) continue;
                acc += (A)x[(((size_t)mj * p.in_h + iy) * p.in_w + ix) * p.minor + mn] *
                       (A)k[(p.kh - 1 - jy) * p.kw + (p.kw - 1 - jx)];
            }
        }
        out[e] = (T)acc;
    }
}

template <typename T, typename A>
int upfirdn2d_any(T* out, const T* x, const T* k, int64_t major, int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y,
                  int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, te_stream_t stream_, const char* what) {
    TE_REQUIRE(out && x && k, TE_ERR_NULL, "%s: out/x/k is NULL", what);
    TE_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0 && kh > 0 && kw > 0, TE_ERR_SHAPE, "%s: bad dims", what);
    TE_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, TE_ERR_SHAPE, "%s: up/down must be > 0", what);
    FirParams p{};
    p.in_h = in_h; p.in_w = in_w;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    TE_REQUIRE(p.out_h > 0 && p.out_w > 0, TE_ERR_SHAPE, "%s: empty output (%d x %d)", what, p.out_h, p.out_w);
    p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.minor = minor; p.major = major;
    if (major == 0) return 0;
    const int64_t total = major * p.out_h * p.out_w * minor;
    const int grid = (int)std::min<int64_t>(te::cdiv(total, 256), te::kNumCU * 16);
    fir_direct_any_kernel<T, A><<<grid, 256, 0, (hipStream_t)stream_>>>(out, x, k, p);
    return te::launch_status(what);
}
}  // namespace

extern "C" int te_upfirdn2d_f16(void* out, const void* x, const void* k, int64_t major, int in_h, int in_w, int minor, int kh,
                                int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                te_stream_t stream) {
    return upfirdn2d_any<__half, float>((__half*)out, (const __half*)x, (const __half*)k, major, in_h, in_w, minor, kh, kw, up_x,
                                        up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream, "te_upfirdn2d_f16");
}

extern "C" int te_upfirdn2d_f64(double* out, const double* x, const double* k, int64_t major, int in_h, int in_w, int minor,
                                int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                int pad_y1, te_stream_t stream) {
    return upfirdn2d_any<double, double>(out, x, k, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                         pad_y0, pad_y1, stream, "te_upfirdn2d_f64");
}

extern "C" int te_upfirdn2d_f32(float* out, const float* x, const float* k, int64_t major, int in_h, int in_w, int minor,
                                int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                int pad_y0, int pad_y1, const float* b, int64_t size_b, int act, float alpha,
                                float scale, te_stream_t stream_) {
    TE_REQUIRE(out && x && k, TE_ERR_NULL, "te_upfirdn2d_f32: out/x/k is NULL");
    TE_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0 && kh > 0 && kw > 0, TE_ERR_SHAPE,
               "te_upfirdn2d_f32: bad dims");
    TE_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, TE_ERR_SHAPE, "te_upfirdn2d_f32: up/down must be > 0");
    TE_REQUIRE(act == 0 || act == 3, TE_ERR_UNSUPPORTED, "te_upfirdn2d_f32: act must be 0 or 3");
    TE_REQUIRE(!(b || act) || minor == 1, TE_ERR_UNSUPPORTED, "te_upfirdn2d_f32: fused epilogue needs minor == 1");
    TE_REQUIRE(!b || size_b > 0, TE_ERR_SHAPE, "te_upfirdn2d_f32: bias given but size_b <= 0");
    FirParams p{};
    p.in_h = in_h; p.in_w = in_w;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    TE_REQUIRE(p.out_h > 0 && p.out_w > 0, TE_ERR_SHAPE, "te_upfirdn2d_f32: empty output (%d x %d)", p.out_h, p.out_w);
    p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.minor = minor; p.major = major;
    p.size_b = (int)size_b; p.act = act; p.alpha = alpha; p.scale = (b || act) ? scale : 1.f;
    if (major == 0) return 0;
    hipStream_t s = (hipStream_t)stream_;
    const bool sq = (up_x == up_y) && (down_x == down_y) && minor == 1;
    if (sq && kh == 4 && kw == 4 && up_x == 1 && down_x == 1 && in_w >= 4 && (int64_t)in_h * in_w * 4 < 0x7FFFFFFF)
        launch_blur44<0>(out, x, k, b, p, s, nullptr, nullptr);
    else if (sq && kh == 4 && kw == 4 && up_x == 2 && down_x == 1) launch_tile<2, 1, 4, 4>(out, x, k, b, p, s);
    else if (sq && kh == 4 && kw == 4 && up_x == 1 && down_x == 2) launch_tile<1, 2, 4, 4>(out, x, k, b, p, s);
    else {
        const int64_t total = major * p.out_h * p.out_w * minor;
        const int grid = (int)std::min<int64_t>(te::cdiv(total, 256), te::kNumCU * 16);
        fir_direct_kernel<<<grid, 256, 0, s>>>(out, x, k, b, p);
    }
    return te::launch_status("te_upfirdn2d_f32");
}
