// og_recon.hip -- the reconstruction kernel of 20 ms CELT-only frames with an 8 KB LDS working set per frame (five waves per SIMD).
//
// A translation unit of its own because the working set is ONE __shared__ object (FrameLds, og_state.hpp) whose layout
// is chosen at compile time: here OG_RECON_TIGHT selects the layout without the folding-history rows, the packet buffer and
// the entropy-decoding arrays, with the synthesis buffer starting inside X.  Only the phase-major band loop and the
// synthesis run from this layout; frames it does not take (recon_fast_eligible, og_celt_split.hpp) are left to the general
// kernel k_celt_recon in og_api.hip, which is launched right behind this one and skips the frames done here.
// Why: measured on the 10 KB layout, k_celt_recon's time goes with 1 / (waves per SIMD) -- 2.69 ms at three, 2.09 ms at four.
//
// Round 3: OG_RECON_FRAMES frames per workgroup, one per wave, and ONE leaf pass for all of them.  A frame has ~48 PVQ leaves of
// very different cost (a 176-coefficient leaf with one pulse next to 8-coefficient ones), a wave runs as long as its most
// expensive leaf, and the kernel is bound by vector-instruction issue: decoded one leaf per lane by the frame's own wave the
// leaf pass was 40 % of the kernel's instructions at 19 % lane use.  Here the leaves of the workgroup's frames are pooled,
// ranked by an estimate of their cost (a counting sort over 64 cost classes in LDS) and dealt out in that order: the first wave
// gets the 64 most expensive leaves of ALL frames, the last wave the cheapest (or none), so the sum of the waves' maxima -- what
// the SIMDs issue -- is about half of what the frames' own maxima add up to (tools/leaf_pool_model.py).  A leaf's result goes
// to its own frame's spectrum wherever the lane that decodes it sits (LDS is the workgroup's), and the waves share one copy of
// the PVQ table.  Everything before and after the leaf pass is wave-private as before: sync points are wave-scoped
// (OG_SYNC_WAVE: LDS instructions of a wave execute in order), only the leaf pass has workgroup barriers, and every wave of
// the workgroup reaches them whatever its frame is.
#define OG_RECON_TIGHT 1
#ifndef OG_RECON_FRAMES
#define OG_RECON_FRAMES 1
#endif
#if OG_RECON_FRAMES > 1
#define OG_LANE ((int)(threadIdx.x & 63))
#define OG_WAVE ((int)(threadIdx.x >> 6))
#define OG_SYNC_WAVE 1
#define OG_LIGHT_SYNC 1
#endif
#include <hip/hip_runtime.h>
#include "og_celt_split.hpp"

using namespace og;

#ifndef OG_FAST_WAVES
#define OG_FAST_WAVES 5
#endif
constexpr int RG = OG_RECON_FRAMES;
static_assert(RG == 1 || RG == 2 || RG == 4, "frames per reconstruction workgroup");

#if OG_RECON_FRAMES > 1 && defined(OG_RECON_POOL)
// ---- the pooled leaf pass -----------------------------------------------------------------------------------------------
// While the leaves are decoded nothing else lives behind the spectra: frame 0's rows hold the PVQ table, the other frames'
// rows the pool -- the leaves of one round (leaf 64 r + lane of every frame) in ranked order -- and the ranking's counters.
constexpr int POOL = 64 * RG;
OG_DEV PvqLds &pvq_shared() { return *reinterpret_cast<PvqLds *>(&Sx[0].v[V_NORM]); }
OG_DEV u32 *pool_idx() { return reinterpret_cast<u32 *>(&Sx[1].v[V_NORM]); }
OG_DEV u32 *pool_geom() { return pool_idx() + POOL; }
OG_DEV u32 *pool_aux() { return RG == 4 ? reinterpret_cast<u32 *>(&Sx[2].v[V_NORM]) : pool_geom() + POOL; }
OG_DEV u16 *pool_meta() { return reinterpret_cast<u16 *>(pool_aux() + POOL); } // frame | leaf << 2 | spread << 11
OG_DEV u32 *pool_cnt() { return RG == 4 ? reinterpret_cast<u32 *>(&Sx[3].v[V_NORM]) : reinterpret_cast<u32 *>(pool_meta() + POOL); }
OG_DEV u32 *pool_base() { return pool_cnt() + 64; }
OG_DEV u32 *pool_nl() { return pool_base() + 64; } // [RG] leaves per frame, then [RG] = leaves in the pool this round
static_assert((RG == 4 ? 2 * POOL * 4 : 3 * POOL * 4 + POOL * 2 + (128 + RG + 1) * 4) <= (V_MASK - V_NORM) * 2, "pool rows of frame 1");
static_assert(POOL * 4 + POOL * 2 <= (V_MASK - V_NORM) * 2 && (128 + RG + 1) * 4 <= (V_MASK - V_NORM) * 2, "pool rows of frames 2, 3");

// What a leaf costs the wave that decodes it, in vector instructions, roughly (the rates are the leaf function's: ~50 per plain
// step of the walk, ~200 per pulse where zero runs are skipped, 8 per coefficient scaled, 11 per step of a rotation pass --
// two passes of two sweeps where the leaf is long enough for the second stride, 5 per coefficient of a short-block frame's
// collapse mask).  Only the ORDER it induces matters, and only for speed.
OG_DEV int leaf_cost_class(u32 g) {
    const int n = (int)(g >> 11) & 255, k = (int)(g >> 19) & 255, B = (int)(g >> 27) + 1;
    int c = 8 * n + (B > 1 ? 5 * n : 0);
    if (2 * k < n) c += (n >= 8 * B ? 44 : 22) * n;
    c += n > k ? 200 * k : 50 * n;
    return OG_MIN(c >> 7, 63);
}

// n_leaves / spread: the calling wave's frame (0 leaves: no frame, or one without a leaf pass).  Every wave of the workgroup calls
// this, and they all pass the same barriers.
OG_DEV void leaf_pass_pooled(const ParseRec *rec, int n_leaves, int spread) {
    const int tid = (int)threadIdx.x, lane = OG_LANE, wave = OG_WAVE;
    { // the one PVQ table, loaded by everybody
        PvqLds &T = pvq_shared();
        for (int t = tid; t < ROM_PVQ_RR_LEN; t += 64 * RG) T.rr[t] = rom_pvq_rr[t];
        if (tid < 16) T.rb[tid] = rom_pvq_rb[tid];
    }
    if (lane == 0) pool_nl()[wave] = (u32)n_leaves;
    __syncthreads();
    int most = 0;
    for (int w = 0; w < RG; w++) most = OG_MAX(most, (int)pool_nl()[w]);
    most = OG_UNI(most);
    for (int r0 = 0; r0 < most; r0 += 64) {
        // rank this round's leaves by cost class, most expensive first
        const int t = r0 + lane;
        const bool have = t < n_leaves;
        u32 g = 0;
        int key = 0;
        if (have) {
            g = rec->leaf[t].geom;
            key = 63 - leaf_cost_class(g);
        }
        if (tid < 64) pool_cnt()[tid] = 0;
        __syncthreads();
        u32 mine = 0;
        if (have) mine = atomicAdd(&pool_cnt()[key], 1u);
        __syncthreads();
        if (tid < 64) { // exclusive prefix over the 64 classes by the first wave
            const u32 c = pool_cnt()[tid];
            u32 incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 up = (u32)__shfl_up((int)incl, d, 64);
                if (tid >= d) incl += up;
            }
            pool_base()[tid] = incl - c;
            if (tid == 63) pool_nl()[RG] = incl;
        }
        __syncthreads();
        if (have) {
            const int slot = (int)(pool_base()[key] + mine);
            pool_idx()[slot] = rec->leaf[t].idx;
            pool_geom()[slot] = g;
            pool_aux()[slot] = rec->leaf[t].aux;
            pool_meta()[slot] = (u16)(wave | t << 2 | spread << 11);
        }
        __syncthreads();
        if (tid < (int)pool_nl()[RG]) { // slot tid: the wave's 64 slots are neighbours in the ranking
            const u32 pg = pool_geom()[tid], aux = pool_aux()[tid];
            const int meta = pool_meta()[tid], fr = meta & 3, leaf = (meta >> 2) & 511;
            const u32 cm = pvq_leaf_lane(Sx[fr].v, pvq_shared(), (int)(pg >> 11) & 255, (int)(pg >> 19) & 255, pool_idx()[tid], V_X + (int)(pg & 2047),
                                         (int)(pg >> 27) + 1, (i32)(aux & 0xffff), meta >> 11);
            Sx[fr].leaf_mask_row()[leaf] = (u16)(cm << ((aux >> 16) & 15));
        }
        __syncthreads();
    }
}
#endif

// `hybrid`: the step's hybrid frames come through the split path too (their SILK half by k_silk_parse / k_silk_synth): this
// kernel then also takes their CELT half
__global__ void __launch_bounds__(64 * RG, OG_FAST_WAVES) k_celt_recon_fb(const FrameDesc *__restrict__ descs, StreamState *st,
                                                                           const ParseRec *recs, ReconOut *rout, int n, int n_streams,
                                                                           int hybrid, u32 *started, const LeafOut *leaves) {
    // `leaves`: what k_celt_leaves (og_leaves.hip) decoded of this step's frames -- null: the frame's own wave decodes its leaves
    // `started` (steps queued as a window, opusgpu_decode_steps_device): every 64th workgroup counts itself in when it starts --
    // the de-emphasis of the step before is held until the first round of this launch has its places
    if (started && threadIdx.x == 0 && (blockIdx.x & 63) == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int f = (int)blockIdx.x * RG + OG_WAVE;
    // the wave's frame, if it has one this kernel takes
    StreamState *sp = nullptr;
    const ParseRec *rec = nullptr;
    ReconCtx rx;
    rx.leaves = false;
    rx.flags = 0;
    bool mine = false;
    int pos = 0;
    u32 lg = 0, la = 0, li = 0;
    OG_PROF_INIT();
    if (f < n) {
        // Round trip 1, everything that hangs off the frame's index: its descriptor, the record's header words, the lane's leaf
        // and band-energy entries.  Round trip 2, what hangs off the stream's index: the stream's scalars and energy histories.
        // (The loads are written before anything waits on one of them: the wave pays two memory latencies here, not fifteen.)
        rec = &recs[f];
        const FrameDesc d = descs[f];
        const int lane = OG_LANE;
        const i32 w_rec = reinterpret_cast<const i32 *>(rec)[lane < 16 ? lane : 15];
        {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 lf = *reinterpret_cast<const u32x4 *>(&rec->leaf[lane]);
            li = lf[0];
            lg = lf[1];
            la = lf[2];
        }
        rx.pre_bandE = rec->bandE[lane < 2 * NBANDS ? lane : 0];
        rx.pre_pulses = rec->pulses[lane < NBANDS ? lane : 0];
        const int mode = desc_mode(d.flags);
        bool ok = !(d.stream < 0 || d.stream >= n_streams || !(mode == MODE_CELT || (mode == MODE_HYBRID && hybrid)) || desc_rfc(d.flags));
        if (ok) {
            sp = &st[d.stream];
            const i32 w_st = recon_hdr_fetch(sp, rec); // (lanes 0-15 re-read the record's words: one instruction for all)
            rx.pre_logE1 = sp->celt.logE1[lane < 2 * NBANDS ? lane : 0];
            rx.pre_logE2 = sp->celt.logE2[lane < 2 * NBANDS ? lane : 0];
            rx.pre = true;
            recon_hdr_unpack(lane < 16 ? w_rec : w_st, rx.h);
            if (mode == MODE_HYBRID && (rx.h.flags & RF_SKIP)) ok = false; // the single-kernel path already reported this frame
        }
        if (ok) {
            pos = rx.h.ring_pos; // where the frame's first sample goes
            mine = recon_begin(sp, rec, mode, desc_channels(d.flags), RECON_FAST_ONLY, rx);
        }
    }
#if OG_RECON_FRAMES > 1 && defined(OG_RECON_POOL)
    leaf_pass_pooled(rec, mine && rx.leaves ? rx.h.n_leaves : 0, (int)(rx.flags >> RF_SPREAD_SHIFT) & 3);
#if defined(OG_RABL) && OG_RABL == 1
    return;
#endif
#else
    if (mine && rx.leaves) {
        if (leaves)
            recon_leaves_fetch(rec, rx, &leaves[f], lg, la);
        else {
            pvq_tab_load();
            OG_SYNC();
#if defined(OG_RABL) && OG_RABL == 1
            return;
#endif
            recon_leaves_own(rec, rx, true, lg, la, li);
        }
    }
#endif
    if (mine) {
        const int ret = recon_finish(sp, rec, rx);
        if (OG_LANE == 0) rout[f] = ReconOut{ret, pos};
    }
    OG_PROF_FLUSH();
}

extern "C" void og_launch_celt_recon_fb(hipStream_t s, const void *descs, void *streams, const void *recs, void *rout, int n,
                                        int n_streams, int hybrid, unsigned *started, const void *leaves) {
    hipLaunchKernelGGL(k_celt_recon_fb, dim3((n + RG - 1) / RG), dim3(64 * RG), 0, s, (const FrameDesc *)descs, (StreamState *)streams,
                       (const ParseRec *)recs, (ReconOut *)rout, n, n_streams, hybrid, started, (const LeafOut *)leaves);
}
extern "C" int og_celt_recon_fb_signals(int n) { return ((n + RG - 1) / RG + 63) / 64; }

#ifdef OG_PROF
// profiling builds only: this kernel's section counters (OG_MARK) -- the other translation unit has its own copy
extern "C" int og_recon_fb_prof(unsigned long long *out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
