// orbx_octree.hip -- DistributeOctTree + DivideNode (src/ORBextractor.cc:483-765) on gfx950.
#include "orbx_internal.h"

// -------------------------------------------------------------------------------------------------
// Quadtree distribution.  One workgroup = one (frame, level).  The reference's std::list algorithm
// is restated on arrays:
//   * keys never move: owner[k] is the arena id of the node that currently holds candidate k;
//   * a pass expands a set E of nodes in a processing order pi and the new list is
//       reverse(children of E in creation order) ++ (old list minus E),
//     which is what push_front + erase produce (:619-665, :689-730);
//   * phase A (:596-667): E = every node with > 1 key, pi = list order;
//   * phase B (:675-739): pi = nodes sorted by (size, creation) descending, E = the shortest prefix
//     after which the list has >= N nodes (the reference's break at :732), found with a prefix sum;
//   * the final "best response, first wins" (:746-762) uses the reference's scan order recomputed
//     from (x, y): cells row-major, then rows, then columns.
// The reference's pointer tie-break in the sort (:629,:686) is taken as creation order.
// -------------------------------------------------------------------------------------------------
// Exclusive prefix sum over the workgroup, ONE barrier: every wave scans its lanes in registers (DPP), publishes its total, and
// after the barrier adds up the totals of the waves below it itself (T / 64 <= 16 values, one DPP row scan).  Consecutive calls
// alternate between two sets of totals (wsum[2][16]), so a wave that is one call ahead never overwrites what a slower wave is
// still reading: being two calls ahead needs the barrier of the call in between.
template <int T>
__device__ __forceinline__ int block_excl_scan(int v, int *total, int (*wsum)[16], int &parity)
{
    constexpr int NW = T / 64;
    static_assert(NW <= 16, "one DPP row holds the wave totals");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = orbx_wave_incl_scan(v);
    int *ws = wsum[parity];
    parity ^= 1;
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int w = lane < NW ? ws[lane] : 0;
    w += __builtin_amdgcn_update_dpp(0, w, 0x111, 0xf, 0xf, true);
    w += __builtin_amdgcn_update_dpp(0, w, 0x112, 0xf, 0xf, true);
    w += __builtin_amdgcn_update_dpp(0, w, 0x114, 0xf, 0xf, true);
    w += __builtin_amdgcn_update_dpp(0, w, 0x118, 0xf, 0xf, true);
    *total = __builtin_amdgcn_readlane(w, NW - 1);
    const int below = __builtin_amdgcn_readlane(w, wave > 0 ? wave - 1 : 0);
    return (wave > 0 ? below : 0) + inc - v;
}

template <int T>
struct OctShared {
    int m, prevM, arenaN, lastBase, lastC, nAlive, nE, C, nToExpand, phaseB, done, cutoff, err, firstPass;
    int w_depth, w_passes, w_mid;   // hand-over of the one-wave passes (oct_body): depth reached, whole passes done, stopped after a pass's first half
    int wsum[2][16];
};

__device__ __forceinline__ void oct_child_box(const OrbxNode &p, int q, OrbxNode &c)
{
    const int halfX = (p.x1 - p.x0 + 1) >> 1;   // ceil((UR.x-UL.x)/2), :485
    const int halfY = (p.y1 - p.y0 + 1) >> 1;   // ceil((BR.y-UL.y)/2), :486
    c.x0 = (q & 1) ? p.x0 + halfX : p.x0;
    c.x1 = (q & 1) ? p.x1 : p.x0 + halfX;
    c.y0 = (q & 2) ? p.y0 + halfY : p.y0;
    c.y1 = (q & 2) ? p.y1 : p.y0 + halfY;
}

// Only the nodes created by the previous pass can still be split (every older list node holds one
// key), so the node records live in an LDS window [lastBase, lastBase + lastC) and the key loop looks
// them up there -- no dependent global gather.  One key loop per pass: moving a key to its child and
// counting its quadrant inside that child (the next pass's DivideNode) happen together, with the
// quadrant counters double-buffered.  The keys themselves (x | y << 16 and the owning node id) live in
// REGISTERS for the whole kernel: thread t owns candidates t, t + T, t + 2T, ... (KR of them; 8 or 32 per
// thread, picked by the level's candidate count), so a pass touches no memory but the LDS node window.
// Levels with more than 32 T candidates fall back to owner[] in global memory, read in batches of 8.
// Global memory keeps the candidates and, at the end, the node-id -> list-position map (the arena
// buffer reused as int32).

// -DOCT_TRACE: workgroup (0,0) stamps wall_clock64() (100 MHz) at the section boundaries (tools/dbg/oct_trace.py).
#ifdef OCT_TRACE
__device__ unsigned long long g_oct_trace[256];
#define OCT_T(tag)                                                                              \
    do {                                                                                        \
        __syncthreads();                                                                        \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && tp < 255)                 \
            g_oct_trace[++tp] = ((unsigned long long)(tag) << 56) | (wall_clock64() & 0xFFFFFFFFFFFFFFull); \
    } while (0)
#else
#define OCT_T(tag) do { } while (0)
#endif

// ++ctr[idx] for every lane with idx >= 0, called by whole waves.  Candidates are stored cell by cell, so neighbouring
// keys mostly fall into the same node and quadrant: one LDS atomic per lane would serialise (same address); instead each
// run of equal idx inside the wave adds its length once.
__device__ __forceinline__ void wave_count(uint32_t *ctr, int idx)
{
    const int lane = (int)__lane_id();
    const int prev = orbx_lane_prev(idx);
    const bool head = lane == 0 || prev != idx;
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(head);
    if (head && idx >= 0) {
        const unsigned long long rest = (hm >> lane) >> 1;
        const int len = rest ? __ffsll((long long)rest) : 64 - lane;
        atomicAdd(&ctr[idx], (uint32_t)len);
    }
}

// slots for the splittable nodes of a window: phase A = list order (:602-667; the previous pass pushed its
// children to the front in reverse creation order, the roots are in creation order), phase B = by (size,
// creation) descending (:686-687).  All phase-B keys are distinct, so a node's slot is the number of larger
// keys: counted directly (every thread streams the same LDS words: broadcasts) instead of a bitonic sort.
template <int T>
__device__ __forceinline__ int oct_assign_slots(OrbxNode *ln, int cntN, bool forward, bool phaseB, uint32_t *slotNode,
                                                unsigned long long *scratch, OctShared<T> &sh, int &scanp)
{
    const int tid = threadIdx.x;
    if (!phaseB) {
        int carry = 0;
        for (int base = 0; base < cntN; base += T) {
            const int j = base + tid;
            int cidx = 0, alive = 0;
            if (j < cntN) {
                cidx = forward ? j : cntN - 1 - j;
                alive = ln[cidx].count > 1;
            }
            int tot;
            const int ex = block_excl_scan<T>(alive, &tot, sh.wsum, scanp);
            if (alive) {
                slotNode[carry + ex] = (uint32_t)cidx;
                ln[cidx].slot = carry + ex;
            }
            carry += tot;
        }
        return carry;
    }
    // key = count << 12 | index (count < 2^20: cand_cap; index < 4096: list_cap, checked at create time); the splittable
    // nodes' keys are packed into scratch[] in any order, then ranked
    uint32_t *keys = reinterpret_cast<uint32_t *>(scratch);
    if (tid == 0) sh.nAlive = 0;
    __syncthreads();
    for (int base = 0; base < cntN; base += T) {
        const int j = base + tid;
        const int c = j < cntN ? ln[j].count : 0;
        const unsigned long long am = __builtin_amdgcn_ballot_w64(c > 1);
        int wbase = 0;
        if (__lane_id() == 0 && am) wbase = atomicAdd(&sh.nAlive, __popcll(am));
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        if (c > 1) keys[orbx_prefix_cnt(am, wbase)] = ((uint32_t)c << 12) | (uint32_t)j;
    }
    __syncthreads();
    const int nA = sh.nAlive;
    for (int i = nA + tid; i < ((nA + 3) & ~3); i += T) keys[i] = 0;   // pad to whole uint4 reads
    __syncthreads();
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
    for (int a = tid; a < nA; a += T) {
        const uint32_t key = keys[a];
        int r = 0;
        for (int i = 0; i < (nA + 3) >> 2; i++) {
            const uint4 v = k4[i];
            r += (v.x > key) + (v.y > key) + (v.z > key) + (v.w > key);
        }
        const int j = (int)(key & 0xFFFu);
        slotNode[r] = (uint32_t)j;
        ln[j].slot = r;
    }
    __syncthreads();
    return nA;
}

__device__ __forceinline__ uint32_t oct_quadrant(const OrbxNode &nd, uint32_t xy)
{
    asm volatile("" : "+v"(xy));   // unpack here, every time: hoisting x and y out of the pass loop costs 2 more registers per key
    const int xr = (int)(xy & 0xFFFFu) - ORBX_MINB, yr = (int)(xy >> 16) - ORBX_MINB;
    const int midx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
    return (xr < midx ? 0u : 1u) | (yr < midy ? 0u : 2u);      // :517-527
}

template <int T, int KR, int KS>   // KS: keys per thread that the caller loaded before the candidate count was known (xy0)
__device__ __forceinline__ void oct_body(const uint32_t *xy0, const OrbxPlan &plan, const OrbxWork &wk, OctShared<T> &sh, const OrbxLevel &L,
                                         const int f, const int l, const int n, const int cap, const int N,
                                         uint32_t *cnt0, uint32_t *cnt1, uint32_t *listA, uint32_t *listB, uint32_t *slotNode,
                                         int *childBase, OrbxNode *lnA, OrbxNode *lnB,
                                         const OrbxCand *__restrict__ cand, uint32_t *__restrict__ owner,
                                         int32_t *__restrict__ posOf, OrbxCand *__restrict__ sel,
                                         uint32_t *pathA, uint32_t *pathB, uint32_t *ftCnt, uint32_t *ftId, uint16_t *mapXY, const int D)
{
    // Fast-forward of the early passes (D > 0).  While every splittable node is split, what a pass does to the KEYS is known
    // in advance: a key ends up in the depth-t node whose box contains it, and the boxes are a function of the root box alone
    // (DivideNode halves with ceil, :485-486).  So ONE key loop computes every key's path of D quadrant choices (2 bits per
    // level) and a histogram of the leaves; sums give the key count of every possible node down to depth D (ftCnt); the passes
    // then run on the NODES alone -- slots, children, list order, arena numbering, the termination and phase-B tests are the
    // reference's, fed with counts from the table instead of a key loop per pass -- and record the arena id of every node they
    // create under its path (ftId).  When the tree has reached depth D, or finishes earlier, one key loop turns each key's path
    // into the id of the deepest created node on it (and its quadrant there), which is exactly the state the per-pass key loops
    // would have left.  Paths: g = root << 2t | quadrants, tables indexed fto(t) + g.
    const int nIniF = L.nIni;
    auto fto = [&](int t) { return nIniF * (int)((((1u << (2 * t)) - 1u)) / 3u); };   // entries above depth t
    auto ft_lookup = [&](uint32_t g, int tmax) -> uint32_t {     // deepest created node on the path of leaf g, searched from depth tmax up
        for (int t = tmax; t > 0; t--) {
            const uint32_t id = ftId[fto(t) + (int)(g >> (2 * (D - t)))];
            if (id != 0xFFFFFFFFu) return id;
        }
        return g >> (2 * D);                                    // the root itself (arena id = root index)
    };
    // Key loops.  REG: every key is in wv[]/xyv[] already.  Otherwise each thread first issues the loads of 8 keys
    // (the loop is latency-bound), then processes them.  The trip counts are wave-uniform: wave_count() is wave-wide.
    constexpr bool REG = KR > 0;
    constexpr int KU = REG ? KR : 8;
    uint32_t wv[KU], xyv[KU];
#define KEYLOOP_BEGIN(LD_OWN, LD_XY)                                    \
    for (int kb = 0; kb < n; kb += T * KU) {                            \
        int tidv = tid;                                                 \
        asm volatile("" : "+v"(tidv));   /* recompute k-derived values per loop instead of keeping them live */ \
        if (!REG) {                                                     \
            _Pragma("unroll") for (int u = 0; u < KU; u++) {            \
                const int k = kb + u * T + tidv;                        \
                if (k < n) {                                            \
                    if (LD_OWN) wv[u] = owner[k];                       \
                    if (LD_XY) xyv[u] = cand[k].xy;                     \
                }                                                       \
            }                                                           \
        }                                                               \
        /* Four keys at a time.  The body is straight-line code (predicated by `valid`, LDS reads at clamped indices), so \
           the dependent LDS look-ups of the four keys overlap instead of running one key after the other. */           \
        _Pragma("unroll") for (int u0 = 0; u0 < KU; u0 += 4) {          \
            if (kb + u0 * T >= n) break;                                \
            int aiv[4];                                                 \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {             \
                const int u = u0 + j;                                   \
                const int k = kb + u * T + tidv;                        \
                const bool valid = k < n;                               \
                int ai = -1;      /* counter this key increments, if any */ \
                {
#define SETOWN(v)                                                       \
    do {                                                                \
        const uint32_t nv_ = (v);                                       \
        if (!REG && valid && nv_ != wv[u]) owner[k] = nv_;              \
        wv[u] = nv_;                                                    \
    } while (0)
#define KEYLOOP_END_COUNT(ctr)                                          \
                }                                                       \
                aiv[j] = ai;                                            \
            }                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; j++) wave_count(ctr, aiv[j]);   \
            __builtin_amdgcn_sched_barrier(0);   /* keep the live ranges of 4 keys, not of all KU */ \
        }                                                               \
    }
#define KEYLOOP_END                                                     \
                }                                                       \
                aiv[j] = ai;                                            \
            }                                                           \
            (void)aiv;                                                  \
            __builtin_amdgcn_sched_barrier(0);                          \
        }                                                               \
    }
    static_assert(KU % 4 == 0, "keys are processed four at a time");
    const int tid = threadIdx.x;
    int scanp = 0;   // which set of wave totals the next workgroup scan uses (block_excl_scan)
#ifdef OCT_TRACE
    int tp = 0;
#endif
    OCT_T(0);
    uint32_t *cc = cnt0, *cn = cnt1;          // quadrant counters of this pass (later: child ids) / of the next pass
    if (REG) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int k = u * T + tid;
            xyv[u] = k < n ? (u < KS ? xy0[u < KS ? u : 0] : cand[k].xy) : 0u;
            wv[u] = 0;
        }
    }
    const int nIni = L.nIni;
    const int boxH = L.maxBY - ORBX_MINB;

    // ---- roots (:554-587) ----
    for (int i = tid; i < nIni; i += T) cc[i] = 0;
    if (D > 0) {
        const int E = fto(D + 1);
        for (int i = tid; i < E; i += T) { ftCnt[i] = 0; ftId[i] = 0xFFFFFFFFu; }
    }
    __syncthreads();
    if (D > 0) {
        // A key's path = D quadrant choices.  The x choices depend on x alone (the root and, level after level, which half of the
        // current x range the key is in) and the y choices on y alone, so the box's columns and rows get their D choices ONCE, spread
        // to the even / odd bit positions of the path, and a key's path is two table reads and an OR instead of D rounds of
        // DivideNode arithmetic (:485-528) -- that loop was 20 of the 27 us this section took for the 24 k keys of a 1080p level 0.
        const int offD = fto(D);
        const int boxW = L.maxBX - ORBX_MINB;
        uint16_t *mapX = mapXY, *mapY = mapXY + ((boxW + 7) & ~7);
        for (int xr = tid; xr < boxW; xr += T) {
            int b = (int)__fdiv_rn((float)xr, L.hX);
            b = min(max(b, 0), nIni - 1);
            int x0 = (int)__fmul_rn(L.hX, (float)b), x1 = (int)__fmul_rn(L.hX, (float)(b + 1));
            uint32_t g = (uint32_t)b;
            for (int t = 0; t < D; t++) {
                const int midx = x0 + ((x1 - x0 + 1) >> 1);
                const uint32_t qx = xr < midx ? 0u : 1u;
                x0 = qx ? midx : x0; x1 = qx ? x1 : midx;
                g = (g << 2) | qx;
            }
            mapX[xr] = (uint16_t)g;           // root << 2D | x choices at the even bits (nIni << 2D <= 2800: checked at plan time)
        }
        for (int yr = tid; yr < boxH; yr += T) {
            int y0 = 0, y1 = boxH;
            uint32_t g = 0;
            for (int t = 0; t < D; t++) {
                const int midy = y0 + ((y1 - y0 + 1) >> 1);
                const uint32_t qy = yr < midy ? 0u : 1u;
                y0 = qy ? midy : y0; y1 = qy ? y1 : midy;
                g = (g << 2) | (qy << 1);
            }
            mapY[yr] = (uint16_t)g;           // y choices at the odd bits
        }
        __syncthreads();
        KEYLOOP_BEGIN(false, true)
            const uint32_t xr = min((xyv[u] & 0xFFFFu) - (uint32_t)ORBX_MINB, (uint32_t)(boxW - 1));   // keys lie inside the box; the clamps only
            const uint32_t yr = min((xyv[u] >> 16) - (uint32_t)ORBX_MINB, (uint32_t)(boxH - 1));       // keep a stray coordinate inside the tables
            const uint32_t g = (uint32_t)mapX[xr] | (uint32_t)mapY[yr];
            if (!REG && valid) owner[k] = g;
            wv[u] = g;
            ai = valid ? offD + (int)g : -1;
        KEYLOOP_END_COUNT(ftCnt)
        __syncthreads();
        // key counts of every possible node, leaves up to the roots: levels wider than a wave by the whole workgroup, the
        // rest (<= 64 nodes each) by wave 0 alone, which needs no workgroup barrier between its levels
        int t = D - 1;
        for (; t >= 0 && (nIni << (2 * t)) > 64; t--) {
            const int o = fto(t), o1 = fto(t + 1), cntT = nIni << (2 * t);
            for (int p = tid; p < cntT; p += T) ftCnt[o + p] = ftCnt[o1 + 4 * p] + ftCnt[o1 + 4 * p + 1] + ftCnt[o1 + 4 * p + 2] + ftCnt[o1 + 4 * p + 3];
            __syncthreads();
        }
        if (tid < 64) {
            for (; t >= 0; t--) {
                const int o = fto(t), o1 = fto(t + 1), cntT = nIni << (2 * t);
                if (tid < cntT) ftCnt[o + tid] = ftCnt[o1 + 4 * tid] + ftCnt[o1 + 4 * tid + 1] + ftCnt[o1 + 4 * tid + 2] + ftCnt[o1 + 4 * tid + 3];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            for (int i = tid; i < nIni; i += 64) cc[i] = ftCnt[i];
        }
    } else {
        KEYLOOP_BEGIN(false, true)
            const int xr = (int)(xyv[u] & 0xFFFFu) - ORBX_MINB;
            int b = (int)__fdiv_rn((float)xr, L.hX);
            b = min(max(b, 0), nIni - 1);
            if (!REG && valid) owner[k] = (uint32_t)b;
            wv[u] = (uint32_t)b;
            ai = valid ? b : -1;
        KEYLOOP_END_COUNT(cc)
    }
    __syncthreads();
    for (int i = tid; i < nIni; i += T) {
        OrbxNode nd;
        nd.x0 = (int16_t)(int)__fmul_rn(L.hX, (float)i);
        nd.x1 = (int16_t)(int)__fmul_rn(L.hX, (float)(i + 1));
        nd.y0 = 0;
        nd.y1 = (int16_t)boxH;
        nd.count = (int)cc[i];
        nd.slot = i;
        lnA[i] = nd;
        pathA[i] = (uint32_t)i;
    }
    __syncthreads();
    if (tid == 0) {   // initial list: non-empty roots in order; nIni is small
        int m = 0;
        for (int i = 0; i < nIni; i++)
            if (cc[i] > 0) listA[m++] = (uint32_t)i;
        sh.m = m; sh.done = 0; sh.phaseB = 0; sh.err = 0;
    }
    __syncthreads();
    OCT_T(1);
    uint32_t *cur = listA, *nxt = listB;
    int m = sh.m, arenaN = nIni, lastBase = 0, lastC = nIni, phaseB = 0;
    int depth = 0;                 // depth of the nodes in the window lnA
    bool fast = D > 0;             // the keys still carry their paths (wv = leaf g), not node ids
    int leaf_tmax = -1;            // >= 0: the tree finished while the keys still carried their paths (see the selection)
    // first DivideNode of every splittable root: slots, then the quadrant counts (from the table, or one key loop)
    int nAlive = oct_assign_slots<T>(lnA, lastC, true, false, slotNode, reinterpret_cast<unsigned long long *>(cn), sh, scanp);
    __syncthreads();
    for (int i = tid; i < 4 * nAlive; i += T) cc[i] = 0;
    __syncthreads();
    if (fast) {
        const int o1 = fto(1);
        for (int i = tid; i < nIni; i += T)
            if (lnA[i].count > 1)
                for (int q = 0; q < 4; q++) cc[4 * lnA[i].slot + q] = ftCnt[o1 + 4 * i + q];
    } else {
        KEYLOOP_BEGIN(true, true)
            const uint32_t id = wv[u];
            const OrbxNode nd = lnA[valid ? id : 0u];
            const bool alive = valid && nd.count > 1;
            const uint32_t q = oct_quadrant(nd, xyv[u]);
            ai = alive ? 4 * nd.slot + (int)q : -1;
            SETOWN(alive ? (id | (q << 30)) : id);
        KEYLOOP_END_COUNT(cc)
    }
    __syncthreads();

    OCT_T(2);
    // ---- expansion passes.  Invariant at the top: lnA = node window [lastBase, lastBase+lastC) with slots
    // assigned to its nAlive splittable nodes, cc = their quadrant counts, OWN = node id | quadrant << 30 ----
#define WSYNC_()                                               \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
    while (true) {
        int nE = 0, C = 0;
        bool half_done = false;        // the first half of this pass (children, survivors) was done by the one-wave passes below
        // While the passes run on the table (phase A, the next pass's counts come from the table too) they touch a few hundred node
        // records and no key: as a workgroup pass that is ~12 barrier-separated steps of ~0.3 us for almost no work.  ONE wave
        // runs them instead -- same arrays, same arithmetic, lanes instead of threads, wave scans instead of workgroup scans, no
        // barrier (a wave's LDS instructions execute in order) -- while the other waves wait at one barrier.  It stops after the
        // first half of the first pass it cannot finish alone (the tree is done, phase B begins, or the table ends) and hands the
        // state over through `sh`; the workgroup finishes that pass below.
        if (fast && !phaseB && depth + 2 <= D) {
            if (tid < 64) {
                const int lane = tid;
                int passes = 0, mid = 0;
                for (;;) {
                    // children per slot and their prefix sums (phase A: every splittable node is expanded)
                    int carry = 0;
                    for (int base = 0; base < nAlive; base += 64) {
                        const int s2 = base + lane;
                        int ne = 0;
                        if (s2 < nAlive) ne = (cc[4 * s2] > 0) + (cc[4 * s2 + 1] > 0) + (cc[4 * s2 + 2] > 0) + (cc[4 * s2 + 3] > 0);
                        const int inc = orbx_wave_incl_scan(ne);
                        if (s2 < nAlive) childBase[s2] = carry + inc - ne;
                        carry += __builtin_amdgcn_readlane(inc, 63);
                    }
                    nE = nAlive; C = carry;
                    const int surv = m - nE;
                    if (arenaN + C > L.arena_cap || C + surv > cap) {
                        if (lane == 0) { atomicOr(&wk.errflags[f], (uint32_t)ERRF_TREE_OVERFLOW); wk.nk[f * plan.nlevels + l] = 0; sh.err = 1; }
                        break;
                    }
                    WSYNC_();
                    int myExp = 0;
                    for (int j0 = 0; j0 < 4 * nE; j0 += 64) {
                        const int j = j0 + lane;
                        const bool on = j < 4 * nE;
                        const int s2 = on ? j >> 2 : 0, q = j & 3;
                        const uint4 c4 = *reinterpret_cast<const uint4 *>(&cc[4 * s2]);
                        const int c = (int)(q == 0 ? c4.x : q == 1 ? c4.y : q == 2 ? c4.z : c4.w);
                        const int r = (q > 0 && c4.x > 0) + (q > 1 && c4.y > 0) + (q > 2 && c4.z > 0);
                        WSYNC_();
                        if (on) {
                            if (c > 0) {
                                const uint32_t pi = slotNode[s2];
                                const OrbxNode p = lnA[pi];
                                const int cidx = childBase[s2] + r;
                                OrbxNode ch;
                                oct_child_box(p, q, ch);
                                ch.count = c;
                                ch.slot = cidx;
                                lnB[cidx] = ch;
                                const uint32_t cp = (pathA[pi] << 2) | (uint32_t)q;
                                pathB[cidx] = cp;
                                ftId[fto(depth + 1) + (int)cp] = (uint32_t)(arenaN + cidx);       // depth + 1 <= D here
                                cc[4 * s2 + q] = (uint32_t)(arenaN + cidx);
                                nxt[C - 1 - cidx] = (uint32_t)(arenaN + cidx);
                                myExp += c > 1;
                            } else {
                                cc[4 * s2 + q] = 0xFFFFFFFFu;
                            }
                        }
                    }
                    const int toExpand = __builtin_amdgcn_readlane(orbx_wave_incl_scan(myExp), 63);
                    WSYNC_();
                    int scarry = 0;
                    for (int base = 0; base < m; base += 64) {
                        const int i = base + lane;
                        int keep = 0;
                        uint32_t id = 0;
                        if (i < m) {
                            id = cur[i];
                            keep = 1;
                            if ((int)id >= lastBase) {
                                const OrbxNode nd = lnA[id - lastBase];
                                keep = !(nd.count > 1 && nd.slot < nE);
                            }
                        }
                        const int inc = orbx_wave_incl_scan(keep);
                        if (keep) nxt[C + scarry + inc - 1] = id;
                        scarry += __builtin_amdgcn_readlane(inc, 63);
                    }
                    const int newM2 = C + scarry;
                    const bool done2 = newM2 >= N || newM2 == m;
                    const bool toB = !done2 && newM2 + 3 * toExpand > N;
                    if (done2 || toB) {                        // the workgroup finishes this pass (and needs nToExpand for its own test)
                        if (lane == 0) sh.nToExpand = toExpand;
                        mid = 1;
                        break;
                    }
                    WSYNC_();
                    // slots of the children that split again, in list order = reverse creation order (phase A)
                    int na = 0;
                    for (int base = 0; base < C; base += 64) {
                        const int j = base + lane;
                        int cidx = 0, alive = 0;
                        if (j < C) { cidx = C - 1 - j; alive = lnB[cidx].count > 1; }
                        const int inc = orbx_wave_incl_scan(alive);
                        if (alive) { slotNode[na + inc - 1] = (uint32_t)cidx; lnB[cidx].slot = na + inc - 1; }
                        na += __builtin_amdgcn_readlane(inc, 63);
                    }
                    WSYNC_();
                    // their quadrant counts from the table
                    const int o2 = fto(depth + 2);
                    for (int j = lane; j < C; j += 64)
                        if (lnB[j].count > 1)
                            for (int q = 0; q < 4; q++) cn[4 * lnB[j].slot + q] = ftCnt[o2 + 4 * (int)pathB[j] + q];
                    WSYNC_();
                    m = newM2; lastBase = arenaN; lastC = C; arenaN += C; nAlive = na; depth++; passes++;
                    { uint32_t *t5 = pathA; pathA = pathB; pathB = t5; }
                    { uint32_t *t2 = cur; cur = nxt; nxt = t2; }
                    { OrbxNode *t3 = lnA; lnA = lnB; lnB = t3; }
                    { uint32_t *t4 = cc; cc = cn; cn = t4; }
                    if (depth + 2 > D) break;                  // the next pass's counts are not in the table: the keys come in
                }
                if (lane == 0) {
                    sh.m = m; sh.arenaN = arenaN; sh.lastBase = lastBase; sh.lastC = lastC; sh.nAlive = nAlive; sh.w_depth = depth;
                    sh.w_passes = passes; sh.w_mid = mid; sh.nE = nE; sh.C = C;
                }
            }
            __syncthreads();
            if (sh.err) return;
            if (tid >= 64 && (sh.w_passes & 1)) {              // the other waves follow the buffer swaps of the passes they sat out
                { uint32_t *t5 = pathA; pathA = pathB; pathB = t5; }
                { uint32_t *t2 = cur; cur = nxt; nxt = t2; }
                { OrbxNode *t3 = lnA; lnA = lnB; lnB = t3; }
                { uint32_t *t4 = cc; cc = cn; cn = t4; }
            }
            m = sh.m; arenaN = sh.arenaN; lastBase = sh.lastBase; lastC = sh.lastC; nAlive = sh.nAlive; depth = sh.w_depth;
            half_done = sh.w_mid != 0;
            nE = sh.nE; C = sh.C;
            __syncthreads();                                   // sh is read; the pass below may rewrite it
        }
        int scarry = 0;
        if (!half_done) {
        if (tid == 0) { sh.cutoff = 0x7FFFFFFF; sh.nToExpand = 0; }
        __syncthreads();
        // children per slot, prefix sums, phase-B cut-off (:732)
        int carry = 0;
        for (int base = 0; base < nAlive; base += T) {
            const int s = base + tid;
            int ne = 0;
            if (s < nAlive) ne = (cc[4 * s] > 0) + (cc[4 * s + 1] > 0) + (cc[4 * s + 2] > 0) + (cc[4 * s + 3] > 0);
            int tot;
            const int ex = block_excl_scan<T>(ne, &tot, sh.wsum, scanp);
            if (s < nAlive) {
                childBase[s] = carry + ex;
                if (phaseB && m + carry + ex + ne - (s + 1) >= N) atomicMin(&sh.cutoff, s);
            }
            carry += tot;
        }
        __syncthreads();
        nE = nAlive; C = carry;
        if (phaseB && sh.cutoff != 0x7FFFFFFF) {
            nE = sh.cutoff + 1;
            const int s = nE - 1;
            C = childBase[s] + (cc[4 * s] > 0) + (cc[4 * s + 1] > 0) + (cc[4 * s + 2] > 0) + (cc[4 * s + 3] > 0);
        }
        OCT_T(3);
        const int survivors = m - nE;
        if (arenaN + C > L.arena_cap || C + survivors > cap) {
            if (tid == 0) { atomicOr(&wk.errflags[f], (uint32_t)ERRF_TREE_OVERFLOW); wk.nk[f * plan.nlevels + l] = 0; }
            return;
        }
        // create the children (:489-537) in creation order; cc becomes the child-id table.  One thread per (slot, quadrant):
        // the four lanes of a slot read its four counts (one 16-byte read each), a wave barrier, then each writes its own entry.
        int myExp = 0;
        for (int j0 = 0; j0 < 4 * nE; j0 += T) {
            const int j = j0 + tid;
            const bool on = j < 4 * nE;
            const int s = on ? j >> 2 : 0, q = j & 3;
            const uint4 c4 = *reinterpret_cast<const uint4 *>(&cc[4 * s]);
            const int c = (int)(q == 0 ? c4.x : q == 1 ? c4.y : q == 2 ? c4.z : c4.w);
            const int r = (q > 0 && c4.x > 0) + (q > 1 && c4.y > 0) + (q > 2 && c4.z > 0);   // non-empty quadrants before this one
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (on) {
                if (c > 0) {
                    const uint32_t pi = slotNode[s];
                    const OrbxNode p = lnA[pi];
                    const int cidx = childBase[s] + r;
                    OrbxNode ch;
                    oct_child_box(p, q, ch);
                    ch.count = c;
                    ch.slot = cidx;
                    lnB[cidx] = ch;
                    const uint32_t cp = (pathA[pi] << 2) | (uint32_t)q;
                    pathB[cidx] = cp;
                    if (fast && depth + 1 <= D) ftId[fto(depth + 1) + (int)cp] = (uint32_t)(arenaN + cidx);
                    cc[4 * s + q] = (uint32_t)(arenaN + cidx);
                    nxt[C - 1 - cidx] = (uint32_t)(arenaN + cidx);   // push_front => reversed
                    myExp += c > 1;
                } else {
                    cc[4 * s + q] = 0xFFFFFFFFu;
                }
            }
        }
        if (myExp) atomicAdd(&sh.nToExpand, myExp);
        // survivors keep their relative order behind the new children
        for (int base = 0; base < m; base += T) {
            const int i = base + tid;
            int keep = 0;
            uint32_t id = 0;
            if (i < m) {
                id = cur[i];
                keep = 1;
                if ((int)id >= lastBase) {
                    const OrbxNode nd = lnA[id - lastBase];
                    keep = !(nd.count > 1 && nd.slot < nE);
                }
            }
            int tot;
            const int ex = block_excl_scan<T>(keep, &tot, sh.wsum, scanp);
            if (keep) nxt[C + scarry + ex] = id;
            scarry += tot;
        }
        } else {
            scarry = m - nE;       // the one-wave half pass has written the new list already
        }
        __syncthreads();
        OCT_T(4);
        // termination and phase switch (:671-675, :736)
        const int newM = C + scarry;
        const bool done = newM >= N || newM == m;
        const int phaseBnext = phaseB || (!done && newM + 3 * sh.nToExpand > N);
        __syncthreads();
        if (done && fast) {
            // the keys still carry their paths: the node of a key is the deepest created node on its path
            // -- resolved once per LEAF (a few thousand) instead of once per key, and straight to the node's position in the final
            // list: the selection below reads it from leafPos[path] (the leaf counts' row of the table, dead by now)
            __syncthreads();       // ftId of this pass's children
            leaf_tmax = min(depth + 1, D);
            m = newM;
            { uint32_t *t2 = cur; cur = nxt; nxt = t2; }
            break;
        }
        if (done) {
            // move the keys of expanded nodes to their children; the rest keep their node
            KEYLOOP_BEGIN(true, false)
                const uint32_t w = wv[u];
                const uint32_t id = w & OCT_ID_MASK;
                const bool inwin = valid && (int)id >= lastBase;
                const OrbxNode nd = lnA[inwin ? id - lastBase : 0u];
                const bool alive = inwin && nd.count > 1;
                const bool expd = alive && nd.slot < nE;
                const uint32_t child = cc[expd ? 4 * nd.slot + (w >> 30) : 0u];
                SETOWN(alive ? (expd ? child : id) : w);
            KEYLOOP_END
            m = newM;
            { uint32_t *t2 = cur; cur = nxt; nxt = t2; }
            break;
        }
        // not finished: every splittable node was expanded (a phase-B cut-off always finishes), so all live keys move.
        // Slots for the children, then ONE key loop: key -> child, and its quadrant inside that child.
        const int nAliveNext = oct_assign_slots<T>(lnB, C, false, phaseBnext != 0, slotNode, reinterpret_cast<unsigned long long *>(cn), sh, scanp);
        OCT_T(phaseBnext ? 6 : 5);
        for (int i = tid; i < 4 * nAliveNext; i += T) cn[i] = 0;
        __syncthreads();
        if (fast && depth + 2 <= D) {
            // still ahead of the keys: the quadrant counts of the new children come from the table
            const int o2 = fto(depth + 2);
            for (int j = tid; j < C; j += T)
                if (lnB[j].count > 1)
                    for (int q = 0; q < 4; q++) cn[4 * lnB[j].slot + q] = ftCnt[o2 + 4 * (int)pathB[j] + q];
        } else if (fast) {
            // the table ends here: bring the keys in.  A key's node is the deepest created node on its path; if that is one of
            // the children just made and it will split, count the key's quadrant inside it (the next pass's DivideNode)
            KEYLOOP_BEGIN(true, true)
                const uint32_t id = ft_lookup(wv[u], min(depth + 1, D));
                const bool child = valid && (int)id >= arenaN;
                const OrbxNode ch = lnB[child ? id - (uint32_t)arenaN : 0u];
                const bool deep = child && ch.count > 1;
                const uint32_t q = oct_quadrant(ch, xyv[u]);
                ai = deep ? 4 * ch.slot + (int)q : -1;
                SETOWN(deep ? (id | (q << 30)) : id);
            KEYLOOP_END_COUNT(cn)
            fast = false;
        } else {
        KEYLOOP_BEGIN(true, true)
            const uint32_t w = wv[u];
            const uint32_t id = w & OCT_ID_MASK;
            const bool inwin = valid && (int)id >= lastBase;
            const OrbxNode nd = lnA[inwin ? id - lastBase : 0u];
            const bool expd = inwin && nd.count > 1;                      // every splittable node is expanded here
            const uint32_t child = cc[expd ? 4 * nd.slot + (w >> 30) : 0u];
            const OrbxNode ch = lnB[expd ? child - (uint32_t)arenaN : 0u];
            const bool deep = expd && ch.count > 1;
            const uint32_t q = oct_quadrant(ch, xyv[u]);
            ai = deep ? 4 * ch.slot + (int)q : -1;
            SETOWN(expd ? (deep ? (child | (q << 30)) : child) : w);
        KEYLOOP_END_COUNT(cn)
        }
        __syncthreads();
        OCT_T(7);
        m = newM; lastBase = arenaN; lastC = C; arenaN += C; nAlive = nAliveNext; phaseB = phaseBnext; depth++;
        { uint32_t *t5 = pathA; pathA = pathB; pathB = t5; }
        { uint32_t *t2 = cur; cur = nxt; nxt = t2; }
        { OrbxNode *t3 = lnA; lnA = lnB; lnB = t3; }
        { uint32_t *t4 = cc; cc = cn; cn = t4; }
    }

    // ---- final selection (:743-762) ----
    unsigned long long *best = reinterpret_cast<unsigned long long *>(cn);   // the idle counter buffer
    OCT_T(8);
    for (int i = tid; i < m; i += T) {
        posOf[cur[i]] = i;
        best[i] = 0ull;
    }
    __threadfence_block();
    __syncthreads();
    uint32_t *leafPos = ftCnt + fto(D);
    if (leaf_tmax >= 0) {
        const int nleaf = nIniF << (2 * D);
        for (int g = tid; g < nleaf; g += T) leafPos[g] = (uint32_t)posOf[ft_lookup((uint32_t)g, leaf_tmax)];
        __syncthreads();
    }
    for (int kb = 0; kb < n; kb += T * KU) {
#pragma unroll
        for (int u0 = 0; u0 < KU; u0 += 8) {      // 8 keys at a time: issue the loads, then the updates
            if (kb + u0 * T >= n) break;
            int posv[8];
            uint32_t rv[8], xv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = kb + (u0 + j) * T + tid;
                if (u0 + j < KU && k < n) {
                    const uint32_t own = REG ? wv[(u0 + j) % KU] : owner[k];
                    posv[j] = leaf_tmax >= 0 ? (int)leafPos[own] : posOf[own & OCT_ID_MASK];   // fast finish: own is still the key's path
                    rv[j] = cand[k].resp;
                    xv[j] = REG ? xyv[(u0 + j) % KU] : cand[k].xy;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = kb + (u0 + j) * T + tid;
                if (u0 + j < KU && k < n) {
                    const int xa = (int)(xv[j] & 0xFFFFu) - ORBX_EDGE, ya = (int)(xv[j] >> 16) - ORBX_EDGE;
                    // exact for ya, xa < 2^16 (level sizes are capped there): one multiply instead of an integer division
                    const int cr = L.rcpH ? (int)__umulhi((uint32_t)ya, L.rcpH) : ya, ccol = L.rcpW ? (int)__umulhi((uint32_t)xa, L.rcpW) : xa;
                    const unsigned long long order = ((unsigned long long)cr << 24) | ((unsigned long long)ccol << 12) |
                                                     ((unsigned long long)(ya - __mul24(cr, L.hCell)) << 6) |
                                                     (unsigned long long)(xa - __mul24(ccol, L.wCell));
                    const unsigned long long pack = ((unsigned long long)rv[j] << 56) |
                                                    ((~order & 0xFFFFFFFFFull) << 20) | (unsigned long long)k;
                    atomicMax(&best[posv[j]], pack);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < m; i += T) sel[i] = cand[(int)(best[i] & 0xFFFFFull)];
    if (tid == 0) wk.nk[f * plan.nlevels + l] = (uint32_t)m;
    OCT_T(9);
#ifdef OCT_TRACE
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_oct_trace[0] = (unsigned long long)tp | ((unsigned long long)n << 32);
#endif
#undef KEYLOOP_BEGIN
#undef KEYLOOP_END
#undef KEYLOOP_END_COUNT
#undef SETOWN
}

// One workgroup = one (frame, level).
template <int T>
__global__ __launch_bounds__(T) void k_octree(OrbxPlan plan, OrbxWork wk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char oct_lds[];
    __shared__ OctShared<T> sh;
    const int tid = threadIdx.x;
    // Level-major order (x = frame, y = level): the level-0 workgroups -- the longest chains -- are dispatched first, and a workgroup's XCD is
    // frame mod 8 instead of its LEVEL (with x = level all level-0 chains shared the 32 CUs of XCD 0, two to a CU).
    const int l = blockIdx.y, f = blockIdx.x;
    const OrbxLevel &L = plan.lv[l];
    const int cap = L.list_cap;
    const int N = L.quota;

    uint32_t *cnt0 = reinterpret_cast<uint32_t *>(oct_lds);                // [4*cap]  quadrant counts, then child ids
    uint32_t *cnt1 = cnt0 + 4 * cap;                                       // [4*cap]  the other pass's counters
    uint32_t *listA = cnt1 + 4 * cap;                                      // [cap]
    uint32_t *listB = listA + cap;                                         // [cap]
    uint32_t *slotNode = listB + cap;                                      // [cap]  slot -> index in the node window
    int *childBase = reinterpret_cast<int *>(slotNode + cap);              // [cap]
    OrbxNode *lnA = reinterpret_cast<OrbxNode *>(childBase + cap);         // [cap]  node window
    OrbxNode *lnB = lnA + cap;                                             // [cap]  children of this pass
    uint32_t *pathA = reinterpret_cast<uint32_t *>(lnB + cap);             // [cap]  path (root << 2 depth | quadrants) of the window's nodes
    uint32_t *pathB = pathA + cap;                                         // [cap]  ... of this pass's children
    // fast-forward tables (oct_body): key counts and arena ids of every possible node down to depth fastD.  They sit behind the
    // arrays of the LARGEST list of the launch (the dynamic LDS size is one number per launch).
    uint32_t *ftCnt = reinterpret_cast<uint32_t *>(oct_lds + (size_t)plan.oct_cap_max * 88);
    uint32_t *ftId = ftCnt + plan.oct_ft;
    uint16_t *mapXY = reinterpret_cast<uint16_t *>(ftId + plan.oct_ft);    // [oct_map] path bits per box column, then per box row (oct_body)
    const int D = L.fastD;

    const OrbxCand *cand = wk.cand + (long long)f * plan.cand_frame + L.cand_off;
    uint32_t *owner = wk.owner + (long long)f * plan.cand_frame + L.cand_off;
    int32_t *posOf = reinterpret_cast<int32_t *>(wk.arena + (long long)f * plan.arena_frame + L.arena_off);   // [arena_cap]
    OrbxCand *sel = wk.sel + (long long)f * plan.list_frame + L.list_off;
    // The first keys of every thread are requested before the candidate count has arrived (one memory round trip instead of
    // two at the head of the kernel's critical path); what lies past the count is stale data of an earlier call, never used.
    constexpr int KS = T == 512 ? 12 : 8;
    uint32_t xy0[KS];
#pragma unroll
    for (int u = 0; u < KS; u++) {
        const int k = u * T + tid;
        xy0[u] = k < L.cand_cap ? cand[k].xy : 0u;
    }
    const int n = (int)min(ORBX_CNT(wk, plan, f, l), (uint32_t)L.cand_cap);
    __syncthreads();
    if (tid == 0) {   // self-cleaning: the counter is zero again for the next call (no memset on the hot path)
        wk.ncand[f * plan.nlevels + l] = (uint32_t)n;
        ORBX_CNT(wk, plan, f, l) = 0;
    }
    if (n == 0 || L.nIni <= 0) {
        if (tid == 0) wk.nk[f * plan.nlevels + l] = 0;
        return;
    }
#define OCT_ARGS xy0, plan, wk, sh, L, f, l, n, cap, N, cnt0, cnt1, listA, listB, slotNode, childBase, lnA, lnB, cand, owner, posOf, sel, pathA, pathB, ftCnt, ftId, mapXY, D
    // T = 512 serves the small shapes, many workgroups per CU: stay under 128 VGPRs.  T = 1024 owns its CU anyway.
    if constexpr (T == 512) {
        if (n <= 12 * T)
            oct_body<T, 12, KS>(OCT_ARGS);
        else
            oct_body<T, 0, KS>(OCT_ARGS);
    } else {
        if (n <= 8 * T)
            oct_body<T, 8, KS>(OCT_ARGS);
        else if (n <= 32 * T)
            oct_body<T, 32, KS>(OCT_ARGS);
        else
            oct_body<T, 0, KS>(OCT_ARGS);
    }
#undef OCT_ARGS
}

size_t orbx_octree_lds_bytes(int list_cap_max, int ft_entries, int map_entries) { return (size_t)list_cap_max * 88 + (size_t)ft_entries * 8 + (size_t)map_entries * 2; }

void orbx_launch_octree(const OrbxPlan &plan, const OrbxWork &wk, int nframes, size_t lds_bytes, hipStream_t s)
{
    dim3 grid(nframes, plan.nlevels);
    if (plan.oct_big) {   // 1080p-class levels: tens of thousands of keys per level
        if (lds_bytes > 32 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_octree<1024>, grid, dim3(1024), lds_bytes, s, plan, wk);
    } else {
        if (lds_bytes > 32 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<OCT_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_octree<OCT_THREADS>, grid, dim3(OCT_THREADS), lds_bytes, s, plan, wk);
    }
}

#ifdef OCT_TRACE
extern "C" int orbx_debug_oct_trace(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_trace), sizeof(unsigned long long) * 256);
}
#endif
