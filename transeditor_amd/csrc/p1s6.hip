// 1x1 convolution on the bf16 matrix pipe (round 6, kind TE_CONV_1X1S6): out[b, m, p] = sum_k W[m, k] in[b, k, p] (+ res) - the skip
// branch of the discriminator's ResBlocks (EqualConv2d 1x1 behind the down-sampling FIR, model_spatial_query.py:780-798, :173-181) and
// its data gradient (the same product with the weights transposed).  Until round 6 these launches were the largest group left on the
// fp32 matrix instructions: 3.8 ms of the 119 ms training iteration at 71 - 104 TFLOP/s (profiles/r06_mid_train_kernel_stats.txt).
//
// Arithmetic = wino6.hip's: every fp32 operand is split into three bf16 pieces (h, m, l: 8 + 8 + 8 mantissa bits) and a multiply-add is
// six exact piece products accumulated in fp32, small terms first (mm, hl, lh, hm, mh, hh): fp32-equivalent results (tests/test_gpu_p1s6.py
// holds it to the fp32 kernel's 5e-6 bar against fp64).  No Winograd form exists for a single tap: 6x the algorithmic FLOPs on the bf16 pipe.
//
// A 1x1 convolution has no spatial structure: a sample's H x W plane is a flat run of pixels, so there is no halo, no edge tile, no
// padding - a tile is 256 CONSECUTIVE pixels of the plane (H W % 256 == 0 is required).
//
// Structure = wino6q_kernel (two-image ping-pong form; read wino6.hip's header first):
//   block   512 threads = two groups of four waves; tile = 128 output channels (two weight images of 64) x 256 pixels (two halves of 128)
//   stage   64 input channels = four MFMA K steps; wave (wm, wn) of a group = 32 channels x 64 pixels (two B fragments) per image:
//           4 K steps x 2 fragments x 6 products = 48 MFMAs per multiplying phase, 3 + 6 operand reads per 12 MFMAs
//   LDS     TWO weight images U[k step 4][piece 3][M tile 2][64 lanes][8 bf16] = 24 KB each (image j = 2 s + m lives in buffer j & 1), two half
//           tiles T[piece 3][k step 4][k half 2][128][8 bf16] = 48 KB each: 144 KB.  A weight image is small here, so it is DOUBLE-BUFFERED instead of
//           renewed half by half inside the phase that still reads it (wino6q's protocol, which leaves the LDS-DMA half a phase - less than its
//           issue + L2 latency - and needs a mid-phase barrier): image j + 1 goes to the buffer image j - 1 was read from, half of its slots
//           issued by group 1 at the start of phase 2 j, the other half by group 0 at the start of phase 2 j + 1, each waited for at the END of
//           the issuing group's staging phase - a whole phase for every transfer, ONE barrier per phase
//   phases  group 0:  M0(s) SA M1(s) SB      group 1:  S M0(s) SA M1(s) SB        (one phase behind, the same straight-line loop body)
//           M0 / M1: 48 MFMAs with image 0 / 1 + half of the staging arithmetic of stage s + 1 each (32 of its 64 slots) + the fetch of stage
//           s + 2 behind the last slot that reads an item's registers;  SA: weight DMA;  SB: weight DMA + the split pieces -> T(s + 1)
//           (first version, same box: half-by-half renewal + mid-phase barrier - profiles/experiments/r06_p1s6_check.log)
//   staging ONE item per thread and stage: 8 channels x 4 consecutive pixels (eight 16-byte loads, 512 contiguous bytes per channel and
//           half wave); the item's four pixels end up as four whole 16-byte LDS elements per piece (ds_write_b128: 12 writes per stage).
//           Pixel 4 q + e of the half tile lives at position 32 e + q, so the 32 lanes of a write fill 512 contiguous bytes, and a wave's two
//           B fragments (positions 64 wn + 32 n + lane) are the ADJACENT pixels 4 lane + 2 wn + n: one 8-byte store per accumulator row.
#include "conv_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 512, GT = 256, KS = 64, NKS = 4, BM = 64, TPX = 256, HPX = 128;
constexpr int U_SLOTS = NKS * 3 * 2;                     // 24 fragment slots of 1 KB per weight image; two images in LDS
constexpr int T_CHUNKS = 3 * NKS * 2 * HPX;              // 16-byte chunks of a half tile: 3 072 = 48 KB
constexpr int N_SLOT = 64;                               // 16 units (4 channel pairs x 4 pixels) x 4 steps

struct P1Args {
    float* out; const float* in; const u32x4* U; const float* res;
    int B, K, M, HW, ntiles, mblocks, tiles_per_sample, nt8;
};

__device__ __forceinline__ void p1_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void p1_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(WT, 2) void p1s6_kernel(const P1Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weight image, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wn = wq & 1, gt = tid & (GT - 1);
    u32x4* tl = reinterpret_cast<u32x4*>(smem_raw + 2 * U_SLOTS * 1024) + grp * T_CHUNKS;   // this group's half tile
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mbq = jx % p.mblocks;
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int b = tile / p.tiles_per_sample, p0 = (tile - b * p.tiles_per_sample) * TPX + HPX * grp;    // first pixel of this group's half
    const size_t plane = (size_t)p.HW;
    const float* inb = p.in + (size_t)b * p.K * plane;

    f32x16 acc[2][2];                                    // [image][B fragment]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // staging item of this thread: pixel quad pq (pixels p0 + 4 pq .. + 3), channel octet k8 (channels 8 k8 .. 8 k8 + 7 of the stage)
    const int pq = gt & 31, k8 = gt >> 5;
    const unsigned g_off = (unsigned)(k8 * 8 * p.HW + p0 + 4 * pq);                       // + (stage * 64 + channel-in-octet) * HW
    const int w_chunk = (k8 * HPX) + pq;                 // + piece * NKS * 2 * HPX + e * 32        (k8 == k step * 2 + k half)
    const int MT = p.M >> 5;
    const int nstage = p.K / KS, nimg = 2 * nstage;
    f32x4 rin[8];                                        // [channel of the octet]: four consecutive pixels each
    auto fetch_half = [&](int hh, int s) {               // channels 4 hh .. 4 hh + 3 of the octet (= channel pairs 2 hh, 2 hh + 1)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            rin[4 * hh + c] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KS + 4 * hh + c) * plane + g_off);
    };
    // part `uh` (k steps 2 uh, 2 uh + 1) of image j = 2 s + m into buffer j & 1: 12 fragment slots, 3 per wave of the group
    auto issue_u = [&](int uh, int j) {
        const int mb = 2 * mbq + (j & 1);
        u32x4* ub = ul + (j & 1) * (U_SLOTS * 64);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int jw = wq * 3 + r, ksl = 2 * uh + jw / 6, piece = (jw % 6) >> 1, mt = jw & 1;
            const u32x4* g = p.U + ((size_t)(((j >> 1) * NKS + ksl) * 3 + piece) * MT + 2 * mb + mt) * 64 + (unsigned)lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ub + ((ksl * 3 + piece) * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // ---- the staging arithmetic: 64 slots = 16 units x 4 steps; unit u = channel pair (u >> 2) x pixel (u & 3) of the item
    unsigned res[4][3][4];                               // [pixel][piece][channel pair] = the four dwords of a 16-byte LDS element
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    auto arith = [&](int k) {
        if (k < 0 || k >= N_SLOT) return;
        const int u = k >> 2, j = k & 3, cp = u >> 2, e = u & 3;
        if (j == 0) {
            te = rin[2 * cp][e]; to = rin[2 * cp + 1][e];
            const f32x2 t = {te, to};
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));      // (even, odd) channel in one dword
            res[e][0][cp] = h;
            fe = __builtin_bit_cast(float, h << 16);
            fo = __builtin_bit_cast(float, h & 0xFFFF0000u);
        } else if (j == 1) {
            te -= fe; to -= fo;
        } else if (j == 2) {
            const f32x2 t = {te, to};
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            res[e][1][cp] = m;
            fe = __builtin_bit_cast(float, m << 16);
            fo = __builtin_bit_cast(float, m & 0xFFFF0000u);
        } else {
            te -= fe; to -= fo;
            const f32x2 t = {te, to};
            res[e][2][cp] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            asm volatile("" : "+v"(res[e][2][cp]));
        }
        asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));          // (pin the step where it is written: wino6.hip)
    };
    auto write_res = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                u32x4 v; v[0] = res[e][pc][0]; v[1] = res[e][pc][1]; v[2] = res[e][pc][2]; v[3] = res[e][pc][3];
                tl[w_chunk + pc * (NKS * 2 * HPX) + e * 32] = v;
            }
    };
    const int a_chunk = wm * 64 + lane;                                  // + ((k step * 3 + piece) * 2) * 64 + image buffer
    const int b_chunk = half * HPX + wn * 64 + l31;                      // + (piece * NKS + k step) * 2 * HPX + n * 32

    // one multiplying phase with accumulator set MSET.  Behind the MFMAs: slots 32 MSET .. 32 MSET + 31 of the staging arithmetic (the
    // channel pairs 2 MSET, 2 MSET + 1 of the item, stage s + 1), then the fetch of those channels for stage fs = s + 2.
    auto multiply = [&](auto mset_tag, int fs) {
        constexpr int MSET = decltype(mset_tag)::value;
        bf16x8 av[2][3], bv[2][2][3];
        const u32x4* ua = ul + MSET * (U_SLOTS * 64) + a_chunk;            // (image j = 2 s + MSET: buffer j & 1 == MSET)
        auto rd_a = [&](int ks, int pc) { av[ks & 1][pc] = __builtin_bit_cast(bf16x8, ua[(ks * 3 + pc) * 128]); };
        auto rd_b = [&](int ks, int n, int pc) { bv[ks & 1][n][pc] = __builtin_bit_cast(bf16x8, tl[b_chunk + (pc * NKS + ks) * 2 * HPX + n * 32]); };
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) { rd_a(0, pc); rd_b(0, 0, pc); rd_b(0, 1, pc); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int i = q * 2 + n;                             // MFMA index inside the k step (12)
#ifndef P1_SKIP_MFMA
                    acc[MSET][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ks & 1][PA[q]], bv[ks & 1][n][PB[q]], acc[MSET][n], 0, 0, 0);
#endif
                    if (ks + 1 < NKS && i < 9) {                         // operands of the next k step: 3 + 6 reads
                        if (i < 3) rd_a(ks + 1, i);
                        else rd_b(ks + 1, (i - 3) / 3, (i - 3) % 3);
                    }
#ifndef P1_SKIP_ARITH
                    {
                        const int k = ks * 12 + i;                       // 0 .. 47
                        if (k < 32) arith(32 * MSET + k);
                        if (k == 34) fetch_half(MSET, fs);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        p1_barrier();                                    // end of phase
    };
    // one phase in the staging role.  In phase ph the other group multiplies with image ph >> 1; this group brings in its part (ph & 1:
    // group 1 part 0 in even phases, group 0 part 1 in odd ones) of image (ph >> 1) + 1, whose buffer was last read in phase ph - 1 (ph even)
    // / ph - 2 (ph odd), and waits for it at the end of the phase.  `write`: the split pieces go to the half tile.
    auto stage = [&](int ph, bool write) {
        const int cs = (ph >> 1) + 1;
        if (cs < nimg) issue_u(ph & 1, cs);
        __builtin_amdgcn_sched_barrier(0);
#ifndef P1_SKIP_ARITH
        if (write) write_res();
#endif
        p1_wait_vm();     // (also waits for the fetch of the next stage issued in this group's last multiplying phase: long landed)
        p1_barrier();                                    // end of phase
    };

    // prologue: every group splits and writes its half of stage 0 and fetches stage 1; group 0 brings in the whole weight image 0
    fetch_half(0, 0);
    fetch_half(1, 0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k);
    write_res();
    fetch_half(0, min(1, nstage - 1));
    fetch_half(1, min(1, nstage - 1));
    if (grp == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // (the weight DMA has landed; the 8 loads of stage 1 stay in flight)
    p1_barrier();
    int ph = 0;
    if (grp == 1) { stage(0, false); ph = 1; }
    for (int s = 0; s < nstage; ++s) {
        const int fs = min(s + 2, nstage - 1);
        multiply(std::integral_constant<int, 0>{}, fs);
        stage(ph + 1, false);
        multiply(std::integral_constant<int, 1>{}, fs);
        if (!(grp == 1 && s == nstage - 1)) stage(ph + 3, true);
        ph += 4;
    }

    // epilogue: the two fragments of a wave are the adjacent pixels 4 l31 + 2 wn + {0, 1} of the half tile: one 8-byte store per row
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mbase = (2 * mbq + m) * BM + wm * 32;
        const size_t off0 = ((size_t)b * p.M + mbase) * plane + (size_t)(p0 + 4 * l31 + 2 * wn);
        f32x2 resv[16];
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                resv[r] = *reinterpret_cast<const f32x2*>(p.res + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
            f32x2 v; v[0] = acc[m][0][r]; v[1] = acc[m][1][r];
            if (p.res) { v[0] += resv[r][0]; v[1] += resv[r][1]; }
            *reinterpret_cast<f32x2*>(p.out + off0 + (size_t)dm * plane) = v;
        }
    }
}

}  // namespace

extern "C" int te_conv_p1s6_supported(int B, int K, int M, int H, int W) {
    if (!(B > 0 && K >= KS && K % KS == 0 && M >= 2 * BM && M % (2 * BM) == 0 && H > 0 && W > 0)) return 0;
    const int64_t hw = (int64_t)H * W;
    if (hw % TPX != 0) return 0;
    const int64_t blocks = (int64_t)B * (hw / TPX) * (M / (2 * BM));
    // (a grid that leaves most CUs without a block stays on the fp32 kernel, which splits K instead)
    return ((int64_t)K * hw * 4 < 0x7FFFFFFF && blocks < 0x7FFFFFF0 && blocks >= te::kNumCU / 2) ? 1 : 0;
}

int te_p1s6_launch(float* out, const float* in, const float* U, const float* res, int B, int K, int M, int H, int W, hipStream_t s) {
    TE_REQUIRE(te_conv_p1s6_supported(B, K, M, H, W), TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_1X1S6): needs K %% 64 == 0, M %% 128 == 0, H W %% 256 == 0 (te_conv_p1s6_supported)");
    TE_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(in)) & 15) == 0, TE_ERR_UNSUPPORTED, "te_conv_f32(TE_CONV_1X1S6): 16-byte aligned tensors required");
    P1Args a{};
    a.out = out; a.in = in; a.U = reinterpret_cast<const u32x4*>(U); a.res = res;
    a.B = B; a.K = K; a.M = M; a.HW = H * W;
    a.tiles_per_sample = a.HW / TPX; a.mblocks = M / (2 * BM);
    a.ntiles = B * a.tiles_per_sample;
    a.nt8 = te::xcd_banded() ? (int)te::cdiv(a.ntiles, 8) : 0;
    const int64_t blocks = te::cdiv(a.ntiles, 8) * 8 * a.mblocks;
    const size_t lds = 2 * (size_t)U_SLOTS * 1024 + 2 * (size_t)T_CHUNKS * 16;
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)p1s6_kernel, 160 * 1024);
    p1s6_kernel<<<dim3((unsigned)blocks), WT, lds, s>>>(a);
    return te::launch_status("te_conv_f32(TE_CONV_1X1S6)");
}
