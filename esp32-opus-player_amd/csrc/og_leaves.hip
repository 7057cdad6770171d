// og_leaves.hip -- the PVQ leaves of a step's 20 ms CELT frames, decoded ONE LEAF PER LANE across frames, in order of cost.
//
// alg_unquant (celt.cpp:782) for every leaf the parse kernel recorded: codeword index -> pulses (cwrsi :2545) -> scaled to the
// leaf's gain (normalise_residual :745) -> spreading rotation undone (exp_rotation :707) -> collapse mask (:760).  All of it is
// a serial chain per leaf and independent across leaves; a frame has ~48 leaves whose cost differs by two orders of magnitude (a
// 176-coefficient leaf with one pulse next to 8-coefficient ones), and a wave runs as long as its most expensive lane.  Inside the
// reconstruction kernel -- one frame per wave, that frame's leaves on its lanes -- the leaf pass was 41 % of the kernel's vector
// instructions at 15 % active lanes (profiles/r03): the kernel is bound by vector-instruction issue, so the idle lanes were the
// single largest waste of the step.
//
// Here a workgroup of four waves takes the leaves of LEAVES_FRAMES frames (64 per frame and round), ranks them by an estimate of
// their cost with a counting sort in LDS, and deals them out in that order: the first wave gets the 64 most expensive leaves of
// all the frames, the last wave the cheapest ones (or none) -- and since no frame's later work happens in this kernel, a wave
// that is done simply ends (pooling the leaves inside the reconstruction kernel, og_recon.hip OG_RECON_POOL, was measured
// slower: there the cheap waves wait at a barrier for the expensive one before they can go on with their frames).  Waves of
// similar leaves also diverge less: the walk's three regimes (closed forms for <= 2 pulses, the zero-run search, the dense
// step) mostly run in different waves.
//
// MEASURED (round 3, 65,536 CELT-FB frames, profiles/r03): k_celt_recon_fb without its leaf pass 11.9 k -> 7.3 k vector
// instructions per frame at 74 % active lanes, 1.50 -> 1.12 ms -- but this kernel issues 3.3 k per frame (four or eight frames per
// workgroup alike: a wave still runs the union of the walk's regimes for as long as its slowest lane, and a leaf that is expensive
// in the walk is cheap in the rotation and vice versa) in 0.55 ms, so the in-order step got LONGER (2.57 -> 2.71 ms) and the
// pipelined one much longer (2.16 -> 2.78 ms: one more kernel in the dependency chain, one more LDS tenant).  It is therefore
// OFF by default (og_debug.hpp: OPUSGPU_LEAF_KERNEL=1 turns it on; tests/test_gpu_celt.py keeps it bit-exact).
//
// A leaf is decoded in an LDS arena (its N coefficients at an offset that packs the round's leaves back to back), then copied
// to the frame's LeafOut in HBM at the offset the parse kernel gave it (the running sum of N): the reconstruction kernel
// fetches a frame's packed coefficients with a few wide loads and spreads them over the spectrum.  72 MB per 65,536-frame
// step each way.
#include <hip/hip_runtime.h>
#include "og_celt_split.hpp"

using namespace og;

#ifndef OG_LEAVES_FRAMES
#define OG_LEAVES_FRAMES 4
#endif
#ifndef OG_LEAVES_WAVES
#define OG_LEAVES_WAVES 6 // workgroups per CU the register budget allows (launch bound)
#endif
constexpr int LF = OG_LEAVES_FRAMES;      // frames (= waves) per workgroup
constexpr int LPOOL = 64 * LF;            // leaves per round
constexpr int LARENA = 3008;              // coefficients the arena holds (a round's leaves: 4 x 554 on the bench payloads)
static_assert(LF == 4 || LF == 8, "frames per leaf workgroup");

struct LeafLds {
    PvqLds T;
    u32 idx[LPOOL], geom[LPOOL], aux[LPOOL]; // the round's leaves in ranked order
    u16 meta[LPOOL];                         // frame within the workgroup | leaf index << 3 | spread << 12
    u32 cnt[64], base[64];
    u32 nl[LF], total, total_n, pass_next, wave_n[LF];
    alignas(16) i16 arena[LARENA];
};
__shared__ LeafLds L;

// (what a leaf costs the wave that decodes it, roughly, in vector instructions: only the ORDER it induces matters, and only for speed)
static __device__ __forceinline__ int leaf_cost_class(u32 g) {
    const int n = (int)(g >> 11) & 255, k = (int)(g >> 19) & 255, B = (int)(g >> 27) + 1;
    int c = 8 * n + (B > 1 ? 5 * n : 0);
    if (2 * k < n) c += (n >= 8 * B ? 36 : 18) * n;
    c += n > k ? 160 * k : 50 * n;
    return OG_MIN(c >> 7, 63);
}

__global__ void __launch_bounds__(64 * LF, OG_LEAVES_WAVES * LF / 4) k_celt_leaves(const FrameDesc *__restrict__ descs, const ParseRec *recs, LeafOut *out, int n,
                                                          int n_streams, int hybrid) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = (int)blockIdx.x * LF + wave;
    // the wave's frame: has it leaves the reconstruction kernel of 20 ms frames expects from here?  (k_celt_recon_fb's conditions)
    int n_leaves = 0, spread = 0;
    const ParseRec *rec = nullptr;
    if (f < n) {
        const FrameDesc d = descs[f];
        const int mode = desc_mode(d.flags);
        if (!(d.stream < 0 || d.stream >= n_streams || !(mode == MODE_CELT || (mode == MODE_HYBRID && hybrid)) || desc_rfc(d.flags))) {
            rec = &recs[f];
            ReconHdr h;
            h.flags = (u32)OG_UNI(rec->flags);
            h.n_words = OG_UNI(rec->n_words);
            h.n_leaves = OG_UNI(rec->n_leaves);
            if (recon_fast_eligible(h)) { // (implies neither RF_SKIP nor RF_BAD_CELT)
                n_leaves = h.n_leaves;
                spread = (int)(h.flags >> RF_SPREAD_SHIFT) & 3;
            }
        }
    }
    for (int t = tid; t < ROM_PVQ_RR_LEN; t += 64 * LF) L.T.rr[t] = rom_pvq_rr[t];
    if (tid < 16) L.T.rb[tid] = rom_pvq_rb[tid];
    if (lane == 0) L.nl[wave] = (u32)n_leaves;
    __syncthreads();
    int most = 0;
    for (int w = 0; w < LF; w++) most = OG_MAX(most, (int)L.nl[w]);
    most = OG_UNI(most);
    for (int r0 = 0; r0 < most; r0 += 64) {
        // ---- rank this round's leaves by cost class, most expensive first
        const int t = r0 + lane;
        const bool have = t < n_leaves;
        u32 g = 0;
        int key = 0;
        if (have) {
            g = rec->leaf[t].geom;
            key = 63 - leaf_cost_class(g);
        }
        if (tid < 64) L.cnt[tid] = 0;
        __syncthreads();
        u32 mine = 0;
        if (have) mine = atomicAdd(&L.cnt[key], 1u);
        __syncthreads();
        if (tid < 64) { // exclusive prefix over the 64 classes by the first wave
            const u32 c = L.cnt[tid];
            u32 incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 up = (u32)__shfl_up((int)incl, d, 64);
                if (tid >= d) incl += up;
            }
            L.base[tid] = incl - c;
            if (tid == 63) L.total = incl;
        }
        __syncthreads();
        if (have) {
            const int slot = (int)(L.base[key] + mine);
            L.idx[slot] = rec->leaf[t].idx;
            L.geom[slot] = g;
            L.aux[slot] = rec->leaf[t].aux;
            L.meta[slot] = (u16)(wave | t << 3 | spread << 12);
        }
        __syncthreads();
        // ---- the slot of this thread: its leaf, and where its coefficients go in the arena (the leaves packed in ranked order)
        const int total = (int)L.total;
        const bool work = tid < total;
        const u32 pg = work ? L.geom[tid] : 0u;
        const int pn = (int)(pg >> 11) & 255;
        int incl = pn;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) L.wave_n[wave] = (u32)incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < LF; w++) before += w < wave ? (int)L.wave_n[w] : 0;
        const int my_off = before + incl - pn; // exclusive prefix over the workgroup
        int all_n = 0;
        for (int w = 0; w < LF; w++) all_n += (int)L.wave_n[w];
        // ---- passes: as many leaves as the arena holds at a time (one pass on the bench payloads)
        for (int pass_lo = 0; pass_lo < all_n;) {
            if (tid == 0) L.pass_next = (u32)all_n;
            __syncthreads();
            const bool fits = work && my_off >= pass_lo && my_off + pn <= pass_lo + LARENA;
            if (work && my_off >= pass_lo && !fits) atomicMin(&L.pass_next, (u32)my_off); // the first leaf that has to wait
            for (int z = tid; z < LARENA / 8; z += 64 * LF) *reinterpret_cast<og_v4i *>(&L.arena[8 * z]) = og_v4i{0, 0, 0, 0};
            __syncthreads();
            const int pass_hi = (int)L.pass_next;
            if (fits && my_off < pass_hi) {
                const u32 aux = L.aux[tid];
                const int meta = L.meta[tid], fr = meta & 7, leaf = (meta >> 3) & 511, pos = my_off - pass_lo;
                const u32 cm = pvq_leaf_lane(L.arena, L.T, pn, (int)(pg >> 19) & 255, L.idx[tid], pos, (int)(pg >> 27) + 1, (i32)(aux & 0xffff), meta >> 12);
                LeafOut *lo = &out[(size_t)blockIdx.x * LF + fr];
                lo->mask[leaf] = (u16)(cm << ((aux >> 16) & 15));
                i16 *dst = &lo->coef[aux >> 20];
                for (int j = 0; j < pn; j++) dst[j] = L.arena[pos + j];
            }
            __syncthreads();
            pass_lo = pass_hi;
        }
    }
}

extern "C" void og_launch_celt_leaves(hipStream_t s, const void *descs, const void *recs, void *leaf_out, int n, int n_streams, int hybrid) {
    hipLaunchKernelGGL(k_celt_leaves, dim3((n + LF - 1) / LF), dim3(64 * LF), 0, s, (const FrameDesc *)descs, (const ParseRec *)recs,
                       (LeafOut *)leaf_out, n, n_streams, hybrid);
}
extern "C" size_t og_leaf_out_bytes(void) { return sizeof(LeafOut); }
