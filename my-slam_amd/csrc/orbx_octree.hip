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
template <int T>
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *wsum)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = lane < T / 64 ? wsum[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < T / 64; o <<= 1) {
            const int t = __shfl_up(winc, o);
            if (lane >= o) winc += t;
        }
        if (lane < T / 64) wsum[lane] = winc - w;
        if (lane == T / 64 - 1) wsum[T / 64] = winc;
    }
    __syncthreads();
    const int res = wsum[wave] + inc - v;
    *total = wsum[T / 64];
    __syncthreads();
    return res;
}

template <int T>
struct OctShared {
    int m, prevM, arenaN, lastBase, lastC, nAlive, nE, C, nToExpand, phaseB, done, cutoff, err, firstPass;
    int wsum[T / 64 + 2];
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
// key), so the node records live in an LDS window [lastBase, lastBase + lastC): the two key loops
// of a pass read owner[] and the candidate coordinates with coalesced global loads and look the
// node up in LDS -- no dependent global gather.  Global memory keeps only owner[] and, at the end,
// the node-id -> list-position map (the arena buffer reused as int32).
// The whole distribution of one (frame, level).  LDSKEYS: the keys' coordinates and owners live in LDS
// (levels with up to OCT_KEYCAP candidates), so the key loops never wait on global memory.
#define OCT_KEYCAP 6144
template <int T, bool LDSKEYS>
__device__ __forceinline__ void oct_body(const OrbxPlan &plan, const OrbxWork &wk, OctShared<T> &sh, const OrbxLevel &L,
                                         const int f, const int l, const int n, const int cap, const int N,
                                         uint32_t *cnt, uint32_t *listA, uint32_t *listB, uint32_t *slotNode, int *childBase,
                                         OrbxNode *lnA, OrbxNode *lnB, uint32_t *kxy, uint32_t *kown,
                                         const OrbxCand *__restrict__ cand, uint32_t *__restrict__ owner,
                                         int32_t *__restrict__ posOf, OrbxCand *__restrict__ sel)
{
#define KXY(k) (LDSKEYS ? kxy[(k)] : cand[(k)].xy)
#define OWN(k) (*(LDSKEYS ? &kown[(k)] : &owner[(k)]))
    const int tid = threadIdx.x;
    unsigned long long *sortbuf = reinterpret_cast<unsigned long long *>(cnt);   // alias, phase B
    unsigned long long *best = reinterpret_cast<unsigned long long *>(cnt);      // alias, final
    if (LDSKEYS)
        for (int k = tid; k < n; k += T) kxy[k] = cand[k].xy;
    const int nIni = L.nIni;
    const int boxH = L.maxBY - ORBX_MINB;

    // ---- roots (:554-572) ----
    for (int i = tid; i < nIni; i += T) cnt[i] = 0;
    if (tid == 0) { sh.err = 0; sh.phaseB = 0; sh.done = 0; sh.firstPass = 1; sh.nToExpand = 0; }
    __syncthreads();
    for (int k = tid; k < n; k += T) {
        const int xr = (int)(KXY(k) & 0xFFFFu) - ORBX_MINB;
        int b = (int)__fdiv_rn((float)xr, L.hX);
        b = min(max(b, 0), nIni - 1);
        OWN(k) = (uint32_t)b;
        atomicAdd(&cnt[b], 1u);
    }
    __syncthreads();
    for (int i = tid; i < nIni; i += T) {
        OrbxNode nd;
        nd.x0 = (int16_t)(int)__fmul_rn(L.hX, (float)i);
        nd.x1 = (int16_t)(int)__fmul_rn(L.hX, (float)(i + 1));
        nd.y0 = 0;
        nd.y1 = (int16_t)boxH;
        nd.count = (int)cnt[i];
        nd.slot = i;
        lnA[i] = nd;
    }
    __syncthreads();
    if (tid == 0) {   // initial list: non-empty roots in order (:574-587); nIni is small
        int m = 0;
        for (int i = 0; i < nIni; i++)
            if (cnt[i] > 0) listA[m++] = (uint32_t)i;
        sh.m = m; sh.arenaN = nIni; sh.lastBase = 0; sh.lastC = nIni;
    }
    __syncthreads();
    uint32_t *cur = listA, *nxt = listB;

    // ---- expansion passes ----
    while (true) {
        const int m = sh.m, lastBase = sh.lastBase, lastC = sh.lastC, arenaN = sh.arenaN;
        const int phaseB = sh.phaseB, firstPass = sh.firstPass;
        __syncthreads();
        // S1: slots for the alive nodes (all of them were created by the previous pass)
        int nAlive;
        if (!phaseB) {
            int carry = 0;
            for (int base = 0; base < lastC; base += T) {
                const int j = base + tid;
                int cidx = 0, alive = 0;
                if (j < lastC) {
                    cidx = firstPass ? j : lastC - 1 - j;   // list order of the last children
                    alive = lnA[cidx].count > 1;
                }
                int tot;
                const int ex = block_excl_scan<T>(alive, &tot, sh.wsum);
                if (alive) {
                    slotNode[carry + ex] = (uint32_t)cidx;
                    lnA[cidx].slot = carry + ex;
                }
                carry += tot;
            }
            nAlive = carry;
        } else {
            // order by (size, creation) descending (:686-687).  All keys are distinct, so a node's slot is the
            // number of larger keys: counted directly (every thread streams the same LDS words: broadcasts)
            // instead of log^2 barrier-separated bitonic stages.
            if (tid == 0) sh.nAlive = 0;
            for (int j = tid; j < lastC; j += T) {
                const int c = lnA[j].count;
                sortbuf[j] = c > 1 ? (((unsigned long long)(uint32_t)c << 32) | (uint32_t)j) : 0ull;
            }
            __syncthreads();
            int mine = 0;
            for (int j = tid; j < lastC; j += T) {
                const unsigned long long key = sortbuf[j];
                if (key != 0) {
                    int r = 0;
                    for (int i = 0; i < lastC; i++) r += sortbuf[i] > key;
                    slotNode[r] = (uint32_t)j;
                    lnA[j].slot = r;
                    mine++;
                }
            }
            if (mine) atomicAdd(&sh.nAlive, mine);
            __syncthreads();
            nAlive = sh.nAlive;
        }
        __syncthreads();
        // S2: zero the quadrant counters
        for (int i = tid; i < 4 * nAlive; i += T) cnt[i] = 0;
        if (tid == 0) { sh.cutoff = 0x7FFFFFFF; sh.nToExpand = 0; }
        __syncthreads();
        // S3: quadrant of every key held by an alive node (:513-528)
#pragma unroll 2
        for (int k = tid; k < n; k += T) {
            const uint32_t id = OWN(k);
            if ((int)id >= lastBase) {
                const OrbxNode nd = lnA[id - lastBase];
                if (nd.count > 1) {
                    const uint32_t xy = KXY(k);
                    const int xr = (int)(xy & 0xFFFFu) - ORBX_MINB, yr = (int)(xy >> 16) - ORBX_MINB;
                    const int midx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                    const uint32_t q = (xr < midx ? 0u : 1u) | (yr < midy ? 0u : 2u);
                    atomicAdd(&cnt[4 * nd.slot + q], 1u);
                    OWN(k) = id | (q << 30);
                }
            }
        }
        __syncthreads();
        // S4: children per slot, prefix sums, phase-B cut-off
        int carry = 0;
        for (int base = 0; base < nAlive; base += T) {
            const int s = base + tid;
            int ne = 0;
            if (s < nAlive) ne = (cnt[4 * s] > 0) + (cnt[4 * s + 1] > 0) + (cnt[4 * s + 2] > 0) + (cnt[4 * s + 3] > 0);
            int tot;
            const int ex = block_excl_scan<T>(ne, &tot, sh.wsum);
            if (s < nAlive) {
                childBase[s] = carry + ex;
                if (phaseB && m + carry + ex + ne - (s + 1) >= N) atomicMin(&sh.cutoff, s);
            }
            carry += tot;
        }
        __syncthreads();
        int nE = nAlive, C = carry;
        if (phaseB && sh.cutoff != 0x7FFFFFFF) {
            nE = sh.cutoff + 1;
            const int s = nE - 1;
            C = childBase[s] + (cnt[4 * s] > 0) + (cnt[4 * s + 1] > 0) + (cnt[4 * s + 2] > 0) + (cnt[4 * s + 3] > 0);
        }
        const int survivors = m - nE;
        if (arenaN + C > L.arena_cap || C + survivors > cap) {
            if (tid == 0) { atomicOr(&wk.errflags[f], (uint32_t)ERRF_TREE_OVERFLOW); wk.nk[f * plan.nlevels + l] = 0; }
            return;
        }
        __syncthreads();
        // S5: create the children (:489-537) in creation order; cnt becomes the child-id table
        int myExp = 0;
        for (int s = tid; s < nE; s += T) {
            const OrbxNode p = lnA[slotNode[s]];
            int r = 0;
            for (int q = 0; q < 4; q++) {
                const int c = (int)cnt[4 * s + q];
                if (c > 0) {
                    const int cidx = childBase[s] + r;
                    OrbxNode ch;
                    oct_child_box(p, q, ch);
                    ch.count = c;
                    ch.slot = cidx;
                    lnB[cidx] = ch;
                    cnt[4 * s + q] = (uint32_t)(arenaN + cidx);
                    nxt[C - 1 - cidx] = (uint32_t)(arenaN + cidx);   // push_front => reversed
                    myExp += c > 1;
                    r++;
                } else {
                    cnt[4 * s + q] = 0xFFFFFFFFu;
                }
            }
        }
        if (myExp) atomicAdd(&sh.nToExpand, myExp);
        __syncthreads();
        // S6: move the keys of expanded nodes to their children
#pragma unroll 2
        for (int k = tid; k < n; k += T) {
            const uint32_t w = OWN(k);
            const uint32_t id = w & OCT_ID_MASK;
            if ((int)id >= lastBase) {
                const OrbxNode nd = lnA[id - lastBase];
                if (nd.count > 1) OWN(k) = nd.slot < nE ? cnt[4 * nd.slot + (w >> 30)] : id;
            }
        }
        // S7: survivors keep their relative order behind the new children
        int scarry = 0;
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
            const int ex = block_excl_scan<T>(keep, &tot, sh.wsum);
            if (keep) nxt[C + scarry + ex] = id;
            scarry += tot;
        }
        __syncthreads();
        // S8: bookkeeping + termination (:671-675, :736)
        if (tid == 0) {
            const int newM = C + scarry;
            sh.prevM = m;
            sh.m = newM;
            sh.lastBase = arenaN;
            sh.lastC = C;
            sh.arenaN = arenaN + C;
            sh.firstPass = 0;
            if (newM >= N || newM == m) sh.done = 1;
            else if (!phaseB && newM + 3 * sh.nToExpand > N) sh.phaseB = 1;
        }
        __syncthreads();
        { uint32_t *t2 = cur; cur = nxt; nxt = t2; }
        { OrbxNode *t3 = lnA; lnA = lnB; lnB = t3; }
        if (sh.done) break;
    }

    // ---- final selection (:743-762) ----
    const int m = sh.m;
    __syncthreads();
    for (int i = tid; i < m; i += T) {
        posOf[cur[i]] = i;
        best[i] = 0ull;   // aliases cnt; child ids are no longer needed
    }
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < n; k += T) {
        const uint32_t id = OWN(k) & OCT_ID_MASK;
        const int pos = posOf[id];
        const OrbxCand c = cand[k];
        const int xa = (int)(c.xy & 0xFFFFu) - ORBX_EDGE, ya = (int)(c.xy >> 16) - ORBX_EDGE;
        const int cr = ya / L.hCell, cc = xa / L.wCell;
        const unsigned long long order = ((unsigned long long)cr << 24) | ((unsigned long long)cc << 12) |
                                         ((unsigned long long)(ya - cr * L.hCell) << 6) |
                                         (unsigned long long)(xa - cc * L.wCell);
        const unsigned long long pack = ((unsigned long long)c.resp << 56) |
                                        ((~order & 0xFFFFFFFFFull) << 20) | (unsigned long long)k;
        atomicMax(&best[pos], pack);
    }
    __syncthreads();
    for (int i = tid; i < m; i += T) sel[i] = cand[(int)(best[i] & 0xFFFFFull)];
    if (tid == 0) wk.nk[f * plan.nlevels + l] = (uint32_t)m;
#undef KXY
#undef OWN
}

template <int T>
__global__ __launch_bounds__(T) void k_octree(OrbxPlan plan, OrbxWork wk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char oct_lds[];
    __shared__ OctShared<T> sh;
    const int tid = threadIdx.x;
    const int l = blockIdx.x, f = blockIdx.y;
    const OrbxLevel &L = plan.lv[l];
    const int cap = L.list_cap;
    const int N = L.quota;

    uint32_t *cnt = reinterpret_cast<uint32_t *>(oct_lds);                 // [4*cap]  counts, then child ids
    uint32_t *listA = cnt + 4 * cap;                                       // [cap]
    uint32_t *listB = listA + cap;                                         // [cap]
    uint32_t *slotNode = listB + cap;                                      // [cap]  slot -> index in the node window
    int *childBase = reinterpret_cast<int *>(slotNode + cap);              // [cap]
    OrbxNode *lnA = reinterpret_cast<OrbxNode *>(childBase + cap);         // [cap]  node window (previous pass)
    OrbxNode *lnB = lnA + cap;                                             // [cap]  children of this pass
    uint32_t *kxy = reinterpret_cast<uint32_t *>(lnB + cap);                 // [OCT_KEYCAP] candidate x | y << 16
    uint32_t *kown = kxy + OCT_KEYCAP;                                       // [OCT_KEYCAP] key -> node id

    const OrbxCand *cand = wk.cand + (long long)f * plan.cand_frame + L.cand_off;
    uint32_t *owner = wk.owner + (long long)f * plan.cand_frame + L.cand_off;
    int32_t *posOf = reinterpret_cast<int32_t *>(wk.arena + (long long)f * plan.arena_frame + L.arena_off);   // [arena_cap]
    OrbxCand *sel = wk.sel + (long long)f * plan.list_frame + L.list_off;
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
    if (n <= OCT_KEYCAP)
        oct_body<T, true>(plan, wk, sh, L, f, l, n, cap, N, cnt, listA, listB, slotNode, childBase, lnA, lnB, kxy, kown, cand, owner, posOf, sel);
    else
        oct_body<T, false>(plan, wk, sh, L, f, l, n, cap, N, cnt, listA, listB, slotNode, childBase, lnA, lnB, kxy, kown, cand, owner, posOf, sel);
}

size_t orbx_octree_lds_bytes(int list_cap_max) { return (size_t)list_cap_max * 64 + (size_t)OCT_KEYCAP * 8; }

void orbx_launch_octree(const OrbxPlan &plan, const OrbxWork &wk, int nframes, size_t lds_bytes, hipStream_t s)
{
    dim3 grid(plan.nlevels, nframes);
    int big = 0;
    for (int l = 0; l < plan.nlevels; l++) big = max(big, plan.lv[l].cand_cap);
    if (big >= 100000) {   // 1080p-class levels: tens of thousands of keys per level
        if (lds_bytes > 32 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_octree<1024>, grid, dim3(1024), lds_bytes, s, plan, wk);
    } else {
        if (lds_bytes > 32 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<OCT_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_octree<OCT_THREADS>, grid, dim3(OCT_THREADS), lds_bytes, s, plan, wk);
    }
}
