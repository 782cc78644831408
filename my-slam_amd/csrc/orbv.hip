// orbv.hip -- DBoW2 vocabulary tree: text loader, GPU descent, BowVector / FeatureVector.
// Reference: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259,1338-1424, BowVector.cpp, FeatureVector.cpp,
// ScoringObject.cpp, FORB.cpp of WChen09/My-SLAM.  See include/orbv.h.
#include <cstring>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include "../../include/orbv.h"

static thread_local std::string g_verr;
static int vfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_verr = buf;
    return code;
}
extern "C" const char *orbv_last_error(void) { return g_verr.c_str(); }
#define VHIP(expr)                                                                               \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return vfail(ORBX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct orbv_vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0, device = 0;
    std::vector<int32_t> parent, child_off, child_ids, word_of;   // per node (child lists in push order)
    std::vector<uint8_t> desc;                                   // nnodes x 32
    std::vector<double> weight;
    int nwords = 0;
    // device copies
    int32_t *d_child_off = nullptr, *d_child_ids = nullptr, *d_word_of = nullptr;
    uint8_t *d_desc = nullptr;
    double *d_weight = nullptr;
    // staging
    uint8_t *d_feat = nullptr; int32_t *d_out_i = nullptr; double *d_out_w = nullptr; size_t cap_feat = 0;
    hipStream_t stream = nullptr;
    uint8_t *h_pin = nullptr;      // pinned: [cap_feat * 32] features in, [cap_feat * 16] results out
};

// greedy descent (TemplatedVocabulary.h:1218-1259): one WAVE per feature.  At every node the lanes take one child each
// (all k descriptors are fetched in one memory round trip instead of k dependent ones), the wave's minimum of
// distance << 16 | child position picks the closest child, the first one on ties (strict '<' at :1246).
__global__ __launch_bounds__(256) void k_voc_transform(const int32_t *__restrict__ child_off, const int32_t *__restrict__ child_ids,
                                                      const uint8_t *__restrict__ ndesc, const int32_t *__restrict__ word_of,
                                                      const double *__restrict__ nweight, const uint8_t *__restrict__ feat, int n,
                                                      int nid_level, int32_t *__restrict__ word_id, int32_t *__restrict__ node_id,
                                                      double *__restrict__ weight)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const uint4 *F = reinterpret_cast<const uint4 *>(feat) + 2 * (long long)i;
    const uint4 f0 = F[0], f1 = F[1];
    int final_id = 0, level = 0, nid = 0;
    int c0 = child_off[0], c1 = child_off[1];
    while (c1 > c0) {                                   // !isLeaf()
        ++level;
        uint32_t best = 0xFFFFFFFFu;
        for (int cb = c0; cb < c1; cb += 64) {          // k <= 64 in practice: one trip
            const int c = cb + lane;
            if (c < c1) {
                const uint4 *D = reinterpret_cast<const uint4 *>(ndesc) + 2 * (long long)child_ids[c];
                const uint4 a0 = D[0], a1 = D[1];
                const uint32_t d = __popc(f0.x ^ a0.x) + __popc(f0.y ^ a0.y) + __popc(f0.z ^ a0.z) + __popc(f0.w ^ a0.w) +
                                   __popc(f1.x ^ a1.x) + __popc(f1.y ^ a1.y) + __popc(f1.z ^ a1.z) + __popc(f1.w ^ a1.w);
                best = min(best, (d << 16) | (uint32_t)min(c - c0, 0xFFFF));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
        final_id = child_ids[c0 + (int)(best & 0xFFFFu)];
        if (level == nid_level) nid = final_id;
        c0 = child_off[final_id]; c1 = child_off[final_id + 1];
    }
    if (lane == 0) {
        word_id[i] = word_of[final_id];
        node_id[i] = nid;
        weight[i] = nweight[final_id];
    }
}

extern "C" void orbv_destroy(orbv_vocabulary *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipFree(v->d_child_off); (void)hipFree(v->d_child_ids); (void)hipFree(v->d_word_of); (void)hipFree(v->d_desc);
    (void)hipFree(v->d_weight); (void)hipFree(v->d_feat); (void)hipFree(v->d_out_i); (void)hipFree(v->d_out_w);
    (void)hipHostFree(v->h_pin);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

extern "C" int orbv_load_text(orbv_vocabulary **out, const char *path, int device)
{
    if (!out || !path) return vfail(ORBX_E_INVALID, "NULL argument");
    *out = nullptr;
    std::ifstream f(path);
    if (!f.is_open()) return vfail(ORBX_E_INVALID, "cannot open %s", path);
    std::string s;
    std::getline(f, s);
    std::stringstream ss(s);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    ss >> k >> L >> n1 >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)   // :1358-1362
        return vfail(ORBX_E_INVALID, "%s: not a vocabulary text file (k=%d L=%d scoring=%d weighting=%d)", path, k, L, n1, n2);
    orbv_vocabulary *v = new orbv_vocabulary();
    v->k = k; v->L = L; v->scoring = n1; v->weighting = n2; v->device = device;
    std::vector<std::vector<int32_t>> children(1);
    v->parent.push_back(0); v->word_of.push_back(0); v->weight.push_back(0.0);
    v->desc.assign(32, 0);
    std::vector<uint8_t> is_leaf(1, 0);
    while (std::getline(f, s)) {
        if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;   // see orbv.h: the reference's trailing-line quirk
        std::stringstream sn(s);
        int pid = 0, leaf = 0;
        sn >> pid >> leaf;
        const int nid = (int)v->parent.size();
        if (sn.fail() || pid < 0 || pid >= nid) { delete v; return vfail(ORBX_E_INVALID, "%s: bad node line %d", path, nid); }
        v->parent.push_back(pid);
        children.push_back(std::vector<int32_t>());
        children[pid].push_back(nid);
        for (int i = 0; i < 32; i++) { int e = 0; sn >> e; v->desc.push_back((uint8_t)e); }   // FORB::fromString
        double w = 0;
        sn >> w;
        v->weight.push_back(w);
        is_leaf.push_back(leaf > 0);
        if (leaf > 0) { v->word_of.push_back(v->nwords++); } else v->word_of.push_back(0);
    }
    const int nn = (int)v->parent.size();
    if (children[0].empty()) { delete v; return vfail(ORBX_E_INVALID, "%s: empty vocabulary", path); }
    v->child_off.assign(nn + 1, 0);
    for (int i = 0; i < nn; i++) {
        if (is_leaf[i] && !children[i].empty()) { delete v; return vfail(ORBX_E_INVALID, "%s: leaf %d has children", path, i); }
        v->child_off[i + 1] = v->child_off[i] + (int)children[i].size();
        v->child_ids.insert(v->child_ids.end(), children[i].begin(), children[i].end());
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete v; return vfail(ORBX_E_HIP, "no HIP device: liborbx has no CPU path"); }
    if (device < 0 || device >= ndev) { delete v; return vfail(ORBX_E_INVALID, "device %d of %d", device, ndev); }
#define VALLOC(dst, src)                                                                                       \
    do {                                                                                                       \
        const size_t b_ = (src).size() * sizeof((src)[0]);                                                     \
        if (hipMalloc((void **)&(dst), b_ ? b_ : 256) != hipSuccess ||                                         \
            (b_ && hipMemcpy((dst), (src).data(), b_, hipMemcpyHostToDevice) != hipSuccess)) {                 \
            orbv_destroy(v); return vfail(ORBX_E_HIP, "vocabulary upload failed");                             \
        }                                                                                                      \
    } while (0)
    if (hipSetDevice(device) != hipSuccess) { delete v; return vfail(ORBX_E_HIP, "hipSetDevice failed"); }
    VALLOC(v->d_child_off, v->child_off); VALLOC(v->d_child_ids, v->child_ids); VALLOC(v->d_word_of, v->word_of);
    VALLOC(v->d_desc, v->desc); VALLOC(v->d_weight, v->weight);
#undef VALLOC
    *out = v;
    return ORBX_OK;
}

extern "C" int orbv_info(const orbv_vocabulary *v, int *k, int *L, int *nnodes, int *nwords, int *scoring, int *weighting)
{
    if (!v) return vfail(ORBX_E_INVALID, "NULL handle");
    if (k) *k = v->k; if (L) *L = v->L; if (nnodes) *nnodes = (int)v->parent.size(); if (nwords) *nwords = v->nwords;
    if (scoring) *scoring = v->scoring; if (weighting) *weighting = v->weighting;
    return ORBX_OK;
}

extern "C" int orbv_transform_features(orbv_vocabulary *v, const uint8_t *desc, int n, int levelsup,
                                       int32_t *word_id, int32_t *node_id, double *weight)
{
    if (!v) return vfail(ORBX_E_INVALID, "NULL handle");
    if (n < 0) return vfail(ORBX_E_INVALID, "n < 0");
    if (n == 0) return ORBX_OK;
    if (!desc || !word_id || !node_id || !weight) return vfail(ORBX_E_INVALID, "NULL buffer");
    VHIP(hipSetDevice(v->device));
    if (!v->stream) VHIP(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    if ((size_t)n > v->cap_feat) {
        VHIP(hipStreamSynchronize(v->stream));
        (void)hipFree(v->d_feat); (void)hipFree(v->d_out_i); (void)hipFree(v->d_out_w); (void)hipHostFree(v->h_pin);
        v->d_feat = nullptr; v->d_out_i = nullptr; v->d_out_w = nullptr; v->h_pin = nullptr; v->cap_feat = 0;
        const size_t cap = (size_t)n + (size_t)n / 4 + 64;
        VHIP(hipMalloc((void **)&v->d_feat, cap * 32));
        VHIP(hipMalloc((void **)&v->d_out_i, cap * 16));      // [cap] word | [cap] node | [cap] weight (double): one block, one copy back
        VHIP(hipHostMalloc((void **)&v->h_pin, cap * 48, hipHostMallocDefault));
        v->cap_feat = cap;
    }
    // pinned staging + one stream: one copy in, one kernel, one copy out, one synchronisation
    const size_t cap = v->cap_feat;
    std::memcpy(v->h_pin, desc, (size_t)n * 32);
    VHIP(hipMemcpyAsync(v->d_feat, v->h_pin, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
    const int nid_level = v->L - levelsup;      // <= 0: nid stays 0 (the root), :1228-1229
    int32_t *d_word = v->d_out_i, *d_node = v->d_out_i + cap;
    double *d_w = reinterpret_cast<double *>(v->d_out_i + 2 * cap);
    hipLaunchKernelGGL(k_voc_transform, dim3((n + 3) / 4), dim3(256), 0, v->stream, v->d_child_off, v->d_child_ids, v->d_desc, v->d_word_of,
                       v->d_weight, v->d_feat, n, nid_level, d_word, d_node, d_w);
    VHIP(hipGetLastError());
    uint8_t *h_out = v->h_pin + cap * 32;
    VHIP(hipMemcpyAsync(h_out, v->d_out_i, cap * 16, hipMemcpyDeviceToHost, v->stream));
    VHIP(hipStreamSynchronize(v->stream));
    std::memcpy(word_id, h_out, (size_t)n * 4);
    std::memcpy(node_id, h_out + cap * 4, (size_t)n * 4);
    std::memcpy(weight, h_out + cap * 8, (size_t)n * 8);
    return ORBX_OK;
}

// Keys are (id << 32 | feature index) pushed in ascending feature order, so a *stable* sort on the id alone gives the
// (id, feature index) order of the reference's std::map / push_back.  LSD radix on 11-bit digits of the id: 2000 keys in
// ~15 us where std::sort takes ~75 us.
static void sort_by_id_stable(std::vector<unsigned long long> &key)
{
    const size_t n = key.size();
    if (n < 64) { std::sort(key.begin(), key.end()); return; }
    uint32_t mx = 0;
    for (unsigned long long k : key) mx = std::max(mx, (uint32_t)(k >> 32));
    std::vector<unsigned long long> tmp(n);
    for (int shift = 32; shift < 64 && (mx >> (shift - 32)) != 0; shift += 11) {
        uint32_t cnt[2049] = {0};
        for (size_t i = 0; i < n; i++) cnt[((key[i] >> shift) & 2047u) + 1]++;
        for (int i = 0; i < 2048; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; i++) tmp[cnt[(key[i] >> shift) & 2047u]++] = key[i];
        key.swap(tmp);
    }
}

// TemplatedVocabulary.h:1143-1193 + BowVector.cpp:34-84
extern "C" int orbv_bow_vector(const orbv_vocabulary *v, const int32_t *word_id, const double *weight, int n,
                               int32_t *ids, double *vals, int cap)
{
    if (!v || n < 0 || (n > 0 && (!word_id || !weight))) return vfail(ORBX_E_INVALID, "bad argument");
    const bool must = v->scoring != ORBV_DOT_PRODUCT;                  // ScoringObject.h:74-89
    const bool l2 = v->scoring == ORBV_L2_NORM;
    // The reference's std::map<WordId, WordValue> filled in feature order == features sorted by (word id, feature index)
    // and folded per word in that order (the same floating-point additions in the same order), without a tree node
    // allocation per word.
    std::vector<unsigned long long> key;
    key.reserve((size_t)n);
    for (int i = 0; i < n; i++)
        if (weight[i] > 0) key.push_back(((unsigned long long)(uint32_t)word_id[i] << 32) | (uint32_t)i);
    sort_by_id_stable(key);
    std::vector<int32_t> bid; std::vector<double> bval;
    bid.reserve(key.size()); bval.reserve(key.size());
    const bool tf = v->weighting == ORBV_TF || v->weighting == ORBV_TF_IDF;
    for (size_t a = 0; a < key.size();) {
        const int32_t w = (int32_t)(key[a] >> 32);
        double acc = weight[(uint32_t)key[a]];                         // first occurrence: insert (addWeight / addIfNotExist)
        size_t b = a + 1;
        for (; b < key.size() && (int32_t)(key[b] >> 32) == w; b++)
            if (tf) acc += weight[(uint32_t)key[b]];                   // addWeight: += in feature order; addIfNotExist keeps the first
        bid.push_back(w); bval.push_back(acc);
        a = b;
    }
    if (tf && !bid.empty() && !must) {
        const double nd = (double)bid.size();
        for (double &x : bval) x /= nd;
    }
    if (must) {                                                        // BowVector::normalize
        double norm = 0.0;
        if (!l2) for (double x : bval) norm += fabs(x);
        else { for (double x : bval) norm += x * x; norm = sqrt(norm); }
        if (norm > 0.0) for (double &x : bval) x /= norm;
    }
    if ((int)bid.size() > cap) return vfail(ORBX_E_CAPACITY, "%zu words, capacity %d", bid.size(), cap);
    for (size_t o = 0; o < bid.size(); o++) { ids[o] = bid[o]; vals[o] = bval[o]; }
    return (int)bid.size();
}

// FeatureVector.cpp:31-45 (features of stopped words are not added, TemplatedVocabulary.h:1157-1161)
extern "C" int orbv_feature_vector(const int32_t *node_id, const double *weight, int n,
                                   int32_t *node_ids, int32_t *off, int32_t *idx, int cap_nodes)
{
    if (n < 0 || (n > 0 && (!node_id || !weight)) || !off) return vfail(ORBX_E_INVALID, "bad argument");
    // std::map<NodeId, std::vector<unsigned>> with push_back in feature order == sort by (node id, feature index)
    std::vector<unsigned long long> key;
    key.reserve((size_t)n);
    for (int i = 0; i < n; i++)
        if (weight[i] > 0) key.push_back(((unsigned long long)(uint32_t)node_id[i] << 32) | (uint32_t)i);
    sort_by_id_stable(key);
    int o = 0;
    off[0] = 0;
    for (size_t a = 0; a < key.size(); a++) {
        const int32_t nd = (int32_t)(key[a] >> 32);
        if (a == 0 || nd != (int32_t)(key[a - 1] >> 32)) {
            if (o >= cap_nodes) return vfail(ORBX_E_CAPACITY, "more than %d nodes", cap_nodes);
            node_ids[o++] = nd;
        }
        idx[a] = (int32_t)(uint32_t)key[a];
        off[o] = (int32_t)a + 1;
    }
    return o;
}

// ScoringObject.cpp:23-68
extern "C" double orbv_score_l1(const int32_t *ids1, const double *vals1, int n1, const int32_t *ids2, const double *vals2, int n2)
{
    double score = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (ids1[i] == ids2[j]) { score += fabs(vals1[i] - vals2[j]) - fabs(vals1[i]) - fabs(vals2[j]); i++; j++; }
        else if (ids1[i] < ids2[j]) i++;
        else j++;
    }
    return -score / 2.0;
}
