// EXPERIMENT (round 5, not in the product library): persistent form of the ping-pong split-bf16 Winograd kernel.
// To try it: paste this kernel into csrc/wino6.hip in front of the closing `}  // namespace`, launch it with
//     wino6q_kernel<<<min(blocks, 256), 512, same LDS size as wino6p_kernel, stream>>>(args)
// and select it with a third value of te_conv_wino6_form.
// Result on the MI355X (tools/wino6_ab.py with a third form, same box, batch 16):
//     bit-identical to both product forms at every shape incl. blocks that walk 32 items, every epilogue stage, both layouts;
//     SLOWER: 156 / 182 / 196 / 200 TFLOP/s against 236 / 258 / 275 / 279 of wino6p_kernel.  With the stores of a finished item
//     inside the phase loop the compiler needs 256 VGPRs + 51 - 60 spilled (wino6p_kernel: 193, none) and 98 spilled SGPRs: the 36
//     split results of the staging program are written to scratch behind every slot of the MFMA stream.  Moving the in-loop
//     epilogue to the end of the staging phase and cutting its temporaries from 96 to 24 registers did not remove the spills (the
//     accumulators are redefined - zeroed - in the staging branch, which costs the allocator a second copy at the loop merge).
//     What it would take: the epilogue out of line (an s_setpc call or a separate tail phase with its own register budget), or a
//     zero-C first MFMA per item instead of the zeroing.  Per-block overhead it was after: 14.7 % of an 8-stage launch, 4.1 % of a
//     32-stage one (tools/block_overhead_probe.py).
// Second attempt (the kernel below): the phase body in TWO instances - boundary phases (0 and 2 nstage - 1 of an item) with the
// epilogue / decode code, hot inner phases without (needs #include <type_traits>).  Still bit-identical, still slow: 141 / 177 / 201 /
// 208 TFLOP/s, and already 1.7x slower on launches whose blocks have ONE item - the cost is per phase, not per item.  The function is
// 9 200 lines of ISA (wino6p_kernel: 4 000), i.e. beyond the 64 KB instruction cache two CUs share, while the two waves of every SIMD sit in
// different regions of it (multiplying stream / staging): the suspect is instruction fetch, not the data path.  A persistent form
// needs the item-boundary code OUT of line (real calls for decode and the epilogue) and the prologue's copy of the arithmetic program gone.
// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the PERSISTENT ping-pong form (wino6q_kernel, form 2; header of tools/exp/wino6q_persistent.hip for the first attempt).
// A block keeps its CU and walks a list of (tile, M block) items; the software pipeline continues across the item boundary (while the
// last stages of item k are multiplied, stage 0 of item k + 1 is fetched, transformed and written, its weight halves arrive on the
// usual schedule) and an item's stores happen in its group's next staging phase.  The phase body exists in TWO instances: the
// boundary phases (0 and 2 nstage - 1 of an item) carry the epilogue / decode code, the hot inner phases do not.
__global__ __launch_bounds__(WT, 2) void wino6q_kernel(const Wino6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform by construction: keep the role branches scalar
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + U_CHUNKS * 16) + grp * TP_DWORDS;      // this group's half tile
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    // ---- work items of this (persistent) block: item j of its XCD = (tile of the XCD's band, M block); the block takes j = jx0, jx0 + nbx,
    // jx0 + 2 nbx, ... (nbx blocks per XCD).  Two register sets hold the geometry of the items with even / odd index in that sequence.
    const int xcd = blockIdx.x & 7, jx0 = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int t_lo = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) : xcd, t_hi = p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles;
    const int t_step = p.nt8 ? 1 : 8;
    const int items_x = (t_hi > t_lo ? (t_hi - t_lo + t_step - 1) / t_step : 0) * p.mblocks;
    const int n_items = jx0 < items_x ? (items_x - jx0 + nbx - 1) / nbx : 0;
    if (n_items == 0) return;
    const size_t plane = (size_t)p.H * p.W;
    // uniform (scalar) part of an item's geometry, per parity
    int it_b[2], it_x0[2], it_y0[2], it_mb[2];
    bool it_edge[2];
    // per-thread part: global offsets and edge flags of this thread's three staging items
    unsigned g_off[2][P_IN];
    int e_flag[2][P_IN];
    int l_off[P_IN];
#pragma unroll
    for (int i = 0; i < P_IN; ++i) {
        const int e = gt + GT * i;
        l_off[i] = (((e >> 7) * 2 + (((e >> 4) & 7) >> 2)) * NP + (e & 15)) * 4 + (((e >> 4) & 7) & 3);        // + (piece * 4 + c) * TP_PLANE
    }
    auto decode = [&](int k, int par) {            // item k of this block -> register set `par`
        const int j = jx0 + k * nbx;
        const int tq = j / p.mblocks, mbk = j - tq * p.mblocks;
        const int tile = t_lo + tq * t_step;
        const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, bb = tile / (p.tiles_x * p.tiles_y);
        const int x0 = tx * TW, y0 = ty * TH, yh = y0 + PH * grp;
        const bool edge = (x0 == 0) || (x0 + TW == p.W) || (y0 == 0) || (y0 + TH == p.H);
        unsigned go[P_IN];
        int ef[P_IN];
#pragma unroll
        for (int i = 0; i < P_IN; ++i) {
            const int e = gt + GT * i;
            const int jj = e & 15, q = (e >> 4) & 7, row = e >> 7;
            const int gy = yh - 1 + row;
            const bool left = x0 == 0 && jj == 0, right = x0 + TW == p.W && jj == NP - 1, rowout = gy < 0 || gy >= p.H;
            ef[i] = (left ? 1 : 0) | (right ? 2 : 0) | (rowout ? 4 : 0);
            const int gyc = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
            go[i] = (unsigned)((2 * q * p.H + gyc) * p.W + x0 + 2 * jj - 1 + (left ? 1 : 0) - (right ? 1 : 0));
        }
        if (par) {
            it_b[1] = bb; it_x0[1] = x0; it_y0[1] = y0; it_mb[1] = mbk; it_edge[1] = edge;
#pragma unroll
            for (int i = 0; i < P_IN; ++i) { g_off[1][i] = go[i]; e_flag[1][i] = ef[i]; }
        } else {
            it_b[0] = bb; it_x0[0] = x0; it_y0[0] = y0; it_mb[0] = mbk; it_edge[0] = edge;
#pragma unroll
            for (int i = 0; i < P_IN; ++i) { g_off[0][i] = go[i]; e_flag[0][i] = ef[i]; }
        }
    };
    decode(0, 0);
    if (n_items > 1) decode(1, 1); else decode(0, 1);
    int a_par = 0;                 // parity of the item whose data sits in rin (set when a fetch is issued)

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    const unsigned q2 = 2u * ((gt >> 4) & 7);
    const int MT = p.M >> 5;
    f32x4 rin[P_IN][2];
    float rsc[2] = {1.f, 1.f};     // style scales of this thread's channel pair (the same pair for its three items: 256 % 128 == 0)
    const int nstage = p.K / KC;
    // (uniform base pointer + unsigned 32-bit per-thread offset: the scalar-base addressing form, no 64-bit address registers)
    auto issue = [&](int par, int s) {
        const float* inb = p.in + (size_t)(par ? it_b[1] : it_b[0]) * p.K * plane;
        const float* iscb = p.isc ? p.isc + (size_t)(par ? it_b[1] : it_b[0]) * p.K : nullptr;
        a_par = par;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const float* base = inb + ((size_t)s * KC + h2) * plane;
            if (iscb) rsc[h2] = (iscb + s * KC + h2)[q2];          // (uniform branch around the load: no select, no wait at the join)
#pragma unroll
            for (int i = 0; i < P_IN; ++i) rin[i][h2] = *reinterpret_cast<const f32x4u*>(base + (par ? g_off[1][i] : g_off[0][i]));
        }
    };
    // weight half `uh` of stage s: 36 fragment slots (3 pieces x 6 (tap row, component) groups x 2 M tiles), 9 per wave of the group
    auto issue_u = [&](int uh, int par, int s) {
        const int mb = par ? it_mb[1] : it_mb[0];
        const u32x4* us = p.U + (size_t)s * 36 * MT * 64;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int j = wq * 9 + r, piece = j / 12, rem = j % 12, kc = (rem >> 1) + 6 * uh, mt = rem & 1;
            const int pk = piece * 12 + kc;                                   // == (piece * 3 + ky) * 4 + c
            const u32x4* g = us + ((size_t)pk * MT + 2 * mb + mt) * 64 + (unsigned)lane;      // uniform base + 32-bit lane offset
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ul + (pk * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // ---- the arithmetic of staging (style scale, B^T d, three-piece split) as a PROGRAM OF 51 SLOTS that the multiplying wave runs
    // behind its own MFMAs (one slot per MFMA; all slot indices are compile-time after unrolling).  Measured with the first
    // ping-pong version (profiles/experiments/r05_w6p_phase_profile.log): the same ~50 vector-ALU instructions per item issued by the
    // PARTNER wave of a SIMD that streams MFMAs cost 10 (older wave) to 27 (younger wave) cycles apiece - more than the multiplying
    // phase lasts - while instructions of the multiplying wave itself issue in the shadow of its own MFMAs (32 cycles each).
    //   slots 0..2         : item k: edge patch, style scale (in place in rin)
    //   slots 3 + 4 u + j  : unit u = item * 4 + component, step j:  0: t = (B^T d)_c for the channel pair, h = bf16x2(t)
    //                        1: t -= h    2: m = bf16x2(t), unpack m    3: t -= m, l = bf16x2(t)
    // The results wait in `res` (36 registers) for the staging phase, which only moves them to LDS.
    unsigned res[P_IN][4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    constexpr int N_SLOT = 3 + 4 * 4 * P_IN;
    auto arith = [&](int k) {
        if (k < 0) {
        } else if (k < P_IN) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[k][h2];
                if (a_par ? it_edge[1] : it_edge[0]) {
                    const int f = a_par ? e_flag[1][k] : e_flag[0][k];
                    if (f & 1) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }        // loaded from column 0: element 0 is column -1
                    if (f & 2) { v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = 0.f; }        // loaded one column early: element 3 is column W
                    if (f & 4) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                }
                rin[k][h2] = v * rsc[h2];
                asm volatile("" : "+v"(rin[k][h2]));
            }
        } else if (k < N_SLOT) {
            const int u = (k - P_IN) >> 2, j = (k - P_IN) & 3, i = u >> 2, c = u & 3;
            if (j == 0) {
                const f32x4 e = rin[i][0], o = rin[i][1];                         // even / odd channel of the pair
                te = c == 0 ? e[0] - e[2] : (c == 1 ? e[1] + e[2] : (c == 2 ? e[2] - e[1] : e[1] - e[3]));
                to = c == 0 ? o[0] - o[2] : (c == 1 ? o[1] + o[2] : (c == 2 ? o[2] - o[1] : o[1] - o[3]));
                const f32x2 t = {te, to};
                // v_cvt_pk_bf16_f32 packs (even, odd) into one dword = the LDS element
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[i][c][0] = h;
                fe = __builtin_bit_cast(float, h << 16);
                fo = __builtin_bit_cast(float, h & 0xFFFF0000u);
                asm volatile("" : "+v"(res[i][c][0]));
            } else if (j == 1) {
                te -= fe; to -= fo;
            } else if (j == 2) {
                const f32x2 t = {te, to};
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[i][c][1] = m;
                fe = __builtin_bit_cast(float, m << 16);
                fo = __builtin_bit_cast(float, m & 0xFFFF0000u);
                asm volatile("" : "+v"(res[i][c][1]));
            } else {
                te -= fe; to -= fo;
                const f32x2 t = {te, to};
                res[i][c][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                asm volatile("" : "+v"(res[i][c][2]));
            }
            // (pin the step HERE: the values have no use before the staging phase, and the compiler otherwise sinks the whole
            //  program behind the mid-phase barrier)
            asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));
        }
    };
    auto write_res = [&]() {
#pragma unroll
        for (int i = 0; i < P_IN; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) tl[l_off[i] + (pc * 4 + c) * TP_PLANE] = res[i][c][pc];
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_chunk = ((2 * wrl + rr) * 2 + half) * NP + jj;          // + ((piece * 4 + c) * PR + ky) * 2 * NP       (16-byte chunks)
    const int a_chunk = wm * 64 + lane;                                 // + ((piece * 3 + ky) * 4 + c) * 128

    // ---- epilogue of one item (register set `par`): output transform, demodulation scale, bias, leaky ReLU, residual, mask - in two halves of
    // eight channel rows (the loads of a half are issued up front: inside the store loop every load would wait behind the previous store)
    auto epilogue = [&](int par) {
        const int bb = par ? it_b[1] : it_b[0], x0 = par ? it_x0[1] : it_x0[0], y0 = par ? it_y0[1] : it_y0[0], mb = par ? it_mb[1] : it_mb[0];
        const int wr = grp * 2 + wrl;
        const int mbase = mb * BM + wm * 32;
        // uniform base + 32-bit per-lane offset (a 32-channel slab of the output is at most 32 planes: < 2^31 bytes by te_conv_wino6_supported)
        const size_t ubase = ((size_t)bb * p.M + mbase) * plane;
        const unsigned lane_off = (unsigned)((y0 + 2 * wr + rr) * p.W + x0 + 2 * jj) + (unsigned)(4 * half) * (unsigned)plane;
        const float* oscb = p.osc ? p.osc + (size_t)bb * p.M + mbase : nullptr;
        const float* biasb = p.bias ? p.bias + mbase : nullptr;
        const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            float scv[4], biv[4];
            f32x2 resv[4], mrefv[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = hh * 4 + r4, dm = (r & 3) + 8 * (r >> 2);
                scv[r4] = oscb ? oscb[dm + 4 * half] : 1.f;
                biv[r4] = biasb ? biasb[dm + 4 * half] : 0.f;
                if (p.res) resv[r4] = *reinterpret_cast<const f32x2*>(p.res + ubase + (size_t)(lane_off + (unsigned)dm * (unsigned)plane));
                if (p.mref) mrefv[r4] = *reinterpret_cast<const f32x2*>(p.mref + ubase + (size_t)(lane_off + (unsigned)dm * (unsigned)plane));
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = hh * 4 + r4, dm = (r & 3) + 8 * (r >> 2);
                float v0 = acc[0][r] + acc[1][r] + acc[2][r];
                float v1 = acc[1][r] - acc[2][r] - acc[3][r];
                v0 = v0 * scv[r4] + biv[r4];
                v1 = v1 * scv[r4] + biv[r4];
                if (p.act >= 3) {
                    v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
                    v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
                }
                if (p.res) { v0 += resv[r4][0]; v1 += resv[r4][1]; }
                if (p.mref) {
                    v0 *= mrefv[r4][0] > 0.f ? p.mgain : 0.2f * p.mgain;
                    v1 *= mrefv[r4][1] > 0.f ? p.mgain : 0.2f * p.mgain;
                }
                f32x2 v; v[0] = v0; v[1] = v1;
                *reinterpret_cast<f32x2*>(p.out + ubase + (size_t)(lane_off + (unsigned)dm * (unsigned)plane)) = v;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    };

    // prologue: every group transforms and writes its half of stage 0 of item 0 and fetches stage 1; group 0 brings in the whole weight image
    issue(0, 0);
    if (grp == 0) { issue_u(0, 0, 0); issue_u(1, 0, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k);
    write_res();
    issue(0, 1);
    if (grp == 0) {
        if (p.isc) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    w6p_barrier();
    const int nphase = 2 * nstage;
    // one phase of the pipeline.  EPI (compile time): this instance may end an item - store it, decode the item after next; the
    // instance of the hot inner phases (1 .. 2 nstage - 2) has no such code, so its register allocation is wino6p_kernel's
    auto run_phase = [&](int k, int ph, auto EPI) {
        constexpr bool with_epi = decltype(EPI)::value;
        const bool last = k == n_items - 1 && ph == nphase - 1;
        if ((ph & 1) == grp) {
            // ---- multiply this group's half of stage ph / 2 of item k; behind the MFMAs: the arithmetic of the stage in rin
            bf16x8 av[2][3], bv[2][3];
            auto rd1 = [&](int g, int slot, int q) {
                const int ky = g >> 2, c = g & 3;
                if (q < 3) av[slot][q] = __builtin_bit_cast(bf16x8, ul[a_chunk + ((q * 3 + ky) * 4 + c) * 128]);
                else bv[slot][q - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk + (((q - 3) * 4 + c) * PR + ky) * 2 * NP]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int q = 0; q < 6; ++q) rd1(0, 0, q);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const int slot = g & 1, c = g & 3;
                if (g == 5) w6p_barrier();             // mid-phase barrier
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], acc[c], 0, 0, 0);
                    if (g + 1 < 12 && q < 3) { rd1(g + 1, slot ^ 1, 2 * q); rd1(g + 1, slot ^ 1, 2 * q + 1); }
                    arith(g * 6 + q - W6P_SLOT0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
            // ---- stage.  csl = local stage being staged; csl == nstage means stage 0 of the NEXT item (group 0 in its last phase of
            // an item); group 1 stages stage 0 of item k >= 1 in phase 0 (stage 0 of item 0 was the prologue's)
            const int csl = (ph + 1) >> 1;
            const int it = k + (csl == nstage ? 1 : 0), sl = csl == nstage ? 0 : csl;
            const bool work = it < n_items && !(it == 0 && sl == 0);
            int fit = it, fl = sl + 1;
            if (fl == nstage) { fit = it + 1; fl = 0; }
            const bool fetch = work && fit < n_items;
            if (grp == 1 && work) issue_u(1, it & 1, sl);
            if (work) {
                if (grp == 0 && fetch) issue(fit & 1, fl);
                __builtin_amdgcn_sched_barrier(0);
                write_res();
                if (grp == 1) w6p_wait_vm();
            }
            w6p_barrier();
            if (work && grp == 0) { issue_u(0, it & 1, sl); w6p_wait_vm(); }
            if constexpr (with_epi) {
                // an item just ended for this group (group 0: its last multiplying phase was the previous one of item k; group 1: of item
                // k - 1): store it LATE in the staging phase (the split pieces have left their registers, group 0's fetch and DMA have
                // landed, group 1's fetch is not in flight yet) and decode the item after next into its register set
                const bool epi = grp == 0 ? csl == nstage : (ph == 0 && k >= 1);
                const int ek = grp == 0 ? k : k - 1;
                if (epi) {
                    epilogue(ek & 1);
                    if (ek + 2 < n_items) decode(ek + 2, ek & 1);
                }
            }
            if (fetch && grp == 1) issue(fit & 1, fl);
        }
        if (!last) w6p_barrier();
    };
    // phases 0 and 2 nstage - 1 of every item run the instance WITH the item-boundary code, the phases between them the one without
    {
        int k = 0, ph = 0;
        while (k < n_items) {
            if (ph == 0 || ph == nphase - 1) {
                run_phase(k, ph, std::true_type{});
                if (++ph == nphase) { ph = 0; ++k; }
            } else {
#pragma nounroll
                for (; ph < nphase - 1; ++ph) run_phase(k, ph, std::false_type{});
            }
        }
    }
    // group 1 finished the last item in the last phase (group 0 stored it inside the loop)
    if (grp == 1) epilogue((n_items - 1) & 1);
}

