// orbv.hip -- DBoW2 vocabulary tree: text loader, GPU descent, BowVector / FeatureVector.
// Reference: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259,1338-1424, BowVector.cpp, FeatureVector.cpp,
// ScoringObject.cpp, FORB.cpp of WChen09/My-SLAM.  See include/orbv.h.
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
};

// greedy descent (TemplatedVocabulary.h:1218-1259): one thread per feature
__global__ __launch_bounds__(256) void k_voc_transform(const int32_t *__restrict__ child_off, const int32_t *__restrict__ child_ids,
                                                      const uint8_t *__restrict__ ndesc, const int32_t *__restrict__ word_of,
                                                      const double *__restrict__ nweight, const uint8_t *__restrict__ feat, int n,
                                                      int nid_level, int32_t *__restrict__ word_id, int32_t *__restrict__ node_id,
                                                      double *__restrict__ weight)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 *F = reinterpret_cast<const uint4 *>(feat) + 2 * (long long)i;
    const uint4 f0 = F[0], f1 = F[1];
    int final_id = 0, level = 0, nid = 0;
    do {
        ++level;
        const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
        int best = 0x7FFFFFFF, bid = child_ids[c0];
        for (int c = c0; c < c1; c++) {
            const int id = child_ids[c];
            const uint4 *D = reinterpret_cast<const uint4 *>(ndesc) + 2 * (long long)id;
            const uint4 a0 = D[0], a1 = D[1];
            const int d = __popc(f0.x ^ a0.x) + __popc(f0.y ^ a0.y) + __popc(f0.z ^ a0.z) + __popc(f0.w ^ a0.w) +
                          __popc(f1.x ^ a1.x) + __popc(f1.y ^ a1.y) + __popc(f1.z ^ a1.z) + __popc(f1.w ^ a1.w);
            if (d < best) { best = d; bid = id; }       // strict '<': the first child wins a tie
        }
        final_id = bid;
        if (level == nid_level) nid = final_id;
    } while (child_off[final_id + 1] > child_off[final_id]);   // !isLeaf()
    word_id[i] = word_of[final_id];
    node_id[i] = nid;
    weight[i] = nweight[final_id];
}

extern "C" void orbv_destroy(orbv_vocabulary *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipFree(v->d_child_off); (void)hipFree(v->d_child_ids); (void)hipFree(v->d_word_of); (void)hipFree(v->d_desc);
    (void)hipFree(v->d_weight); (void)hipFree(v->d_feat); (void)hipFree(v->d_out_i); (void)hipFree(v->d_out_w);
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
    if ((size_t)n > v->cap_feat) {
        (void)hipFree(v->d_feat); (void)hipFree(v->d_out_i); (void)hipFree(v->d_out_w);
        v->d_feat = nullptr; v->d_out_i = nullptr; v->d_out_w = nullptr; v->cap_feat = 0;
        VHIP(hipMalloc((void **)&v->d_feat, (size_t)n * 32));
        VHIP(hipMalloc((void **)&v->d_out_i, (size_t)n * 8));
        VHIP(hipMalloc((void **)&v->d_out_w, (size_t)n * 8));
        v->cap_feat = (size_t)n;
    }
    VHIP(hipMemcpy(v->d_feat, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    const int nid_level = v->L - levelsup;      // <= 0: nid stays 0 (the root), :1228-1229
    hipLaunchKernelGGL(k_voc_transform, dim3((n + 255) / 256), dim3(256), 0, 0, v->d_child_off, v->d_child_ids, v->d_desc, v->d_word_of,
                       v->d_weight, v->d_feat, n, nid_level, v->d_out_i, v->d_out_i + n, v->d_out_w);
    VHIP(hipGetLastError());
    VHIP(hipMemcpy(word_id, v->d_out_i, (size_t)n * 4, hipMemcpyDeviceToHost));
    VHIP(hipMemcpy(node_id, v->d_out_i + n, (size_t)n * 4, hipMemcpyDeviceToHost));
    VHIP(hipMemcpy(weight, v->d_out_w, (size_t)n * 8, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// TemplatedVocabulary.h:1143-1193 + BowVector.cpp:34-84
extern "C" int orbv_bow_vector(const orbv_vocabulary *v, const int32_t *word_id, const double *weight, int n,
                               int32_t *ids, double *vals, int cap)
{
    if (!v || n < 0 || (n > 0 && (!word_id || !weight))) return vfail(ORBX_E_INVALID, "bad argument");
    const bool must = v->scoring != ORBV_DOT_PRODUCT;                  // ScoringObject.h:74-89
    const bool l2 = v->scoring == ORBV_L2_NORM;
    std::map<int32_t, double> bow;
    if (v->weighting == ORBV_TF || v->weighting == ORBV_TF_IDF) {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) bow[word_id[i]] += weight[i];           // addWeight
        if (!bow.empty() && !must) {
            const double nd = (double)bow.size();
            for (auto &kv : bow) kv.second /= nd;
        }
    } else {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) bow.insert(std::make_pair(word_id[i], weight[i]));   // addIfNotExist
    }
    if (must) {                                                        // BowVector::normalize
        double norm = 0.0;
        if (!l2) for (auto &kv : bow) norm += fabs(kv.second);
        else { for (auto &kv : bow) norm += kv.second * kv.second; norm = sqrt(norm); }
        if (norm > 0.0) for (auto &kv : bow) kv.second /= norm;
    }
    if ((int)bow.size() > cap) return vfail(ORBX_E_CAPACITY, "%zu words, capacity %d", bow.size(), cap);
    int o = 0;
    for (auto &kv : bow) { ids[o] = kv.first; vals[o] = kv.second; o++; }
    return o;
}

// FeatureVector.cpp:31-45 (features of stopped words are not added, TemplatedVocabulary.h:1157-1161)
extern "C" int orbv_feature_vector(const int32_t *node_id, const double *weight, int n,
                                   int32_t *node_ids, int32_t *off, int32_t *idx, int cap_nodes)
{
    if (n < 0 || (n > 0 && (!node_id || !weight)) || !off) return vfail(ORBX_E_INVALID, "bad argument");
    std::map<int32_t, std::vector<int32_t>> fv;
    for (int i = 0; i < n; i++)
        if (weight[i] > 0) fv[node_id[i]].push_back(i);
    if ((int)fv.size() > cap_nodes) return vfail(ORBX_E_CAPACITY, "%zu nodes, capacity %d", fv.size(), cap_nodes);
    int o = 0, e = 0;
    off[0] = 0;
    for (auto &kv : fv) {
        node_ids[o] = kv.first;
        for (int32_t i : kv.second) idx[e++] = i;
        off[++o] = e;
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
