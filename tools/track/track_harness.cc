// track_harness.cc -- BASELINE config 5's per-frame loop, host side in C++ through the C ABI exactly as INTEGRATION.md
// wires it into the reference:
//     Frame::ExtractORB                 -> orbx_extract_end(frame k) + orbx_extract_begin(frame k + 1) on a second handle
//     Frame::AssignFeaturesToGrid       -> orbm_grid_build
//     Frame::ComputeBoW                 -> orbv_transform_features + orbv_bow_vector + orbv_feature_vector
//     Tracking::TrackReferenceKeyFrame  -> orbm_search_by_bow            (ORBmatcher(0.7, true), src/Tracking.cc:774-779)
//     Tracking::TrackWithMotionModel    -> with the pose stages: orbm_search_by_projection_last (ORBmatcher::SearchByProjection(
//                                          CurrentFrame, LastFrame, 15, mono), src/Tracking.cc:879-883); without them the same
//                                          windows around the previous positions: orbm_search_area_best2 + orbm_rot_filter
//     Optimizer::PoseOptimization       -> orbp_pose_optimization        (src/Tracking.cc:783-787: start from the last pose)
//     Relocalization's PnPsolver        -> orbp_pnp_* (0.99,10,300,4,0.5,5.991; iterate(5)) + orbp_pose_optimization
//                                          (src/Tracking.cc:1392-1445), run every 8th frame to time it, followed by
//                                          orbm_project_points + orbm_search_by_projection_kf (SearchByProjection(
//                                          mCurrentFrame, pKF, sFound, 10, 100), src/Tracking.cc:1459)
// The pose stages need a scene with depth: prep_inputs.py writes the layered stream of my_slam_amd.synth.stream_layers
// and its per-pixel layer map; the "MapPoints" of the previous frame are its keypoints back-projected to the depth of
// their layer through the *estimated* previous pose, so the reported translation error is the drift of the whole chain.
// Inputs: a raw frame file (nframes x H x W bytes), a vocabulary text file and (optional) the layer map, all written by
// tools/track/prep_inputs.py.
// build: g++ -O2 -std=c++17 -I include tools/track/track_harness.cc -L my-slam_amd/lib -lorbx -Wl,-rpath,$PWD/my-slam_amd/lib -o tools/track/track_harness
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "orbx.h"
#include "orbm.h"
#include "orbv.h"
#include "orbp.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHK(e) do { int rc_ = (e); if (rc_ < 0) { fprintf(stderr, "%s failed: %d (%s | %s | %s)\n", #e, rc_, orbx_last_error(), orbm_last_error(), orbv_last_error()); return 2; } } while (0)

// ORBX_TRACK_DUMP=<dir>: every frame's inputs and match tables are written to <dir>/frame_<k>.bin so that a test can run the
// oracle on the SAME extracted features and poses and compare the tables exactly (tests/test_track_harness_gpu.py)
struct Dump {
    FILE *f = nullptr;
    bool open(const char *dir, int k) { char p[1024]; snprintf(p, sizeof p, "%s/frame_%03d.bin", dir, k); f = fopen(p, "wb"); return f != nullptr; }
    template <class T> void put(const T *p, size_t n) { if (f && n) fwrite(p, sizeof(T), n, f); }
    void i32(int32_t v) { put(&v, 1); }
    void close() { if (f) fclose(f); f = nullptr; }
};

struct FrameData {
    std::vector<orbx_keypoint> kps;
    std::vector<uint8_t> desc;
    int n = 0;
    std::vector<int32_t> fv_node, fv_off, fv_idx; int fv_n = 0;
};

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s frames.raw W H nframes voc.txt nfeatures [levelsup [layer.raw baseline s0 s1 s2]]\n", argv[0]); return 1; }
    const char *fpath = argv[1]; const int W = atoi(argv[2]), H = atoi(argv[3]), K = atoi(argv[4]);
    const char *vpath = argv[5]; const int NF = atoi(argv[6]); const int levelsup = argc > 7 ? atoi(argv[7]) : 2;
    const bool pose = argc > 12;
    const char *dump_dir = getenv("ORBX_TRACK_DUMP");
    // The tiles of synth.stream_layers are fixed in IMAGE space: a corner made by the junction of two layers at a tile edge does not
    // move at all from frame to frame, yet it is given the finite depth of its layer.  Such features pull every pose estimate
    // towards "no motion" -- a bias, not noise: the chained error grows linearly along the direction of travel.  With a margin
    // (in level-0 pixels per unit of the feature's scale) they are left out of the pose inputs and the drift shows its true size.
    const float edge_margin = getenv("ORBX_TRACK_EDGE_MARGIN") ? (float)atof(getenv("ORBX_TRACK_EDGE_MARGIN")) : 0.f;
    const float fx = 718.856f, fy = 718.856f, cx = 607.1928f, cy = 185.2157f;      // Examples/Monocular/KITTI00-02.yaml
    std::vector<uint8_t> layer;
    double base = 0, depth[3] = {0, 0, 0};
    if (pose) {
        layer.resize((size_t)W * H);
        FILE *lf = fopen(argv[8], "rb");
        if (!lf || fread(layer.data(), 1, layer.size(), lf) != layer.size()) { fprintf(stderr, "cannot read %s\n", argv[8]); return 1; }
        fclose(lf);
        base = atof(argv[9]);
        for (int r = 0; r < 3; r++) depth[r] = fx * base / atof(argv[10 + r]);
    }
    std::vector<uint8_t> frames((size_t)W * H * K);
    FILE *f = fopen(fpath, "rb");
    if (!f || fread(frames.data(), 1, frames.size(), f) != frames.size()) { fprintf(stderr, "cannot read %s\n", fpath); return 1; }
    fclose(f);

    // two extractor handles: frame k + 1 is staged and enqueued (orbx_extract_begin) as soon as frame k has been collected,
    // so its extraction runs on the GPU underneath the host's matching and pose work on frame k
    orbx_extractor *exs[2] = {nullptr, nullptr}; orbm_matcher *mt = nullptr; orbv_vocabulary *voc = nullptr;
    CHK(orbx_create(&exs[0], NF, 1.2f, 8, 20, 7, 0, W, H, 1));
    CHK(orbx_create(&exs[1], NF, 1.2f, 8, 20, 7, 0, W, H, 1));
    const int cap = orbx_capacity(exs[0]);
    CHK(orbm_create(&mt, 0, cap, cap, 1 << 21));
    CHK(orbv_load_text(&voc, vpath, 0));

    FrameData cur, prev;
    std::vector<int32_t> word(cap), node(cap), bow_ids(cap), match_f(cap), bi(cap), bd(cap), sd(cap), m12(cap), mn(cap), mx(cap);
    std::vector<double> weight(cap), bow_vals(cap);
    std::vector<float> qx(cap), qy(cap), qr(cap), aq(cap), at(cap);
    std::vector<double> t_stage[5];     // per-stage samples; medians are reported (one slow frame must not skew a stage)
    std::vector<double> t_all, t_pose, t_reloc;
    long nm_bow = 0, nm_proj = 0, n_inl = 0, n_reloc = 0, n_reloc_ok = 0, n_reloc_add = 0;
    float Tprev[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};       // frame 0 = world
    std::vector<float> obs(2 * cap), is2(cap), xw(3 * cap), s2(cap);
    std::vector<uint8_t> outl(cap), inl(cap);
    double max_err = 0, max_reloc_err = 0, max_axis[3] = {0, 0, 0}, ck_err[8];
    int ck_frame[8], n_ck = 0;
    float Tlast[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<float> xw_all(3 * cap);
    std::vector<uint8_t> ones(cap, 1);
    std::vector<int32_t> one_obs(cap, 1), cur_obs(cap), cur_match(cap);
    float sfac[8];
    sfac[0] = 1.f;
    for (int l = 1; l < 8; l++) sfac[l] = sfac[l - 1] * 1.2f;
    for (int k = 0; k < K; k++) {
        cur.kps.resize(cap); cur.desc.resize((size_t)cap * 32);
        cur.fv_node.resize(cap); cur.fv_off.resize(cap + 1); cur.fv_idx.resize(cap);
        if (k == 0) CHK(orbx_extract_begin(exs[0], frames.data(), W, H, W));
        const double t0 = now_ms();
        CHK(orbx_extract_end(exs[k & 1], cur.kps.data(), cur.desc.data(), cap, &cur.n));
        if (k + 1 < K) CHK(orbx_extract_begin(exs[(k + 1) & 1], frames.data() + (size_t)(k + 1) * W * H, W, H, W));
        const double t1 = now_ms();
        CHK(orbm_grid_build(mt, cur.kps.data(), cur.n, 0.f, (float)W, 0.f, (float)H));
        const double t2 = now_ms();
        CHK(orbv_transform_features(voc, cur.desc.data(), cur.n, levelsup, word.data(), node.data(), weight.data()));
        const int nb = orbv_bow_vector(voc, word.data(), weight.data(), cur.n, bow_ids.data(), bow_vals.data(), cap);
        cur.fv_n = orbv_feature_vector(node.data(), weight.data(), cur.n, cur.fv_node.data(), cur.fv_off.data(), cur.fv_idx.data(), cap);
        if (nb < 0 || cur.fv_n < 0) { fprintf(stderr, "bow failed: %s\n", orbv_last_error()); return 2; }
        const double t3 = now_ms();
        if (k > 0) {
            int nm = 0;      // previous frame in the role of the reference key frame: every feature has a MapPoint
            CHK(orbm_search_by_bow(mt, prev.desc.data(), prev.kps.data(), prev.n, nullptr, prev.fv_node.data(), prev.fv_off.data(),
                                   prev.fv_idx.data(), prev.fv_n, cur.desc.data(), cur.kps.data(), cur.n, cur.fv_node.data(),
                                   cur.fv_off.data(), cur.fv_idx.data(), cur.fv_n, 0.7f, 1, match_f.data(), &nm));
            nm_bow += nm;
            const double t4 = now_ms();
            Dump dump;
            if (dump_dir && dump.open(dump_dir, k)) {
                const int32_t hdr[8] = {prev.n, cur.n, prev.fv_n, prev.fv_off[prev.fv_n], cur.fv_n, cur.fv_off[cur.fv_n], pose ? 1 : 0, nm};
                dump.put(hdr, 8);
                dump.put(prev.kps.data(), (size_t)prev.n); dump.put(prev.desc.data(), (size_t)prev.n * 32);
                dump.put(cur.kps.data(), (size_t)cur.n); dump.put(cur.desc.data(), (size_t)cur.n * 32);
                dump.put(prev.fv_node.data(), (size_t)prev.fv_n); dump.put(prev.fv_off.data(), (size_t)prev.fv_n + 1); dump.put(prev.fv_idx.data(), (size_t)prev.fv_off[prev.fv_n]);
                dump.put(cur.fv_node.data(), (size_t)cur.fv_n); dump.put(cur.fv_off.data(), (size_t)cur.fv_n + 1); dump.put(cur.fv_idx.data(), (size_t)cur.fv_off[cur.fv_n]);
                dump.put(match_f.data(), (size_t)cur.n);
            }
            if (pose) {
                // MapPoints of the previous frame: Xc = K^-1 (u, v, 1) * depth(layer), Xw = Rprev^T (Xc - tprev)
                int nc = 0;
                for (int i = 0; i < cur.n; i++) {
                    const int j = match_f[i];
                    if (j < 0) continue;
                    const orbx_keypoint &p = prev.kps[j], &c = cur.kps[i];
                    if (edge_margin > 0) {      // ORBX_TRACK_EDGE_MARGIN: leave out features whose patch touches a tile edge of the synthetic scene
                        const float m = edge_margin * sfac[p.octave];
                        const float dx = fmodf(p.x, 96.f), dy = fmodf(p.y, 96.f);
                        if (std::min(dx, 96.f - dx) < m || std::min(dy, 96.f - dy) < m) continue;
                    }
                    const int px = std::min(std::max((int)lrintf(p.x), 0), W - 1), py = std::min(std::max((int)lrintf(p.y), 0), H - 1);
                    const double Z = depth[layer[(size_t)py * W + px]];
                    const double Xc[3] = {(p.x - cx) * Z / fx - Tprev[3], (p.y - cy) * Z / fy - Tprev[7], Z - Tprev[11]};
                    for (int a = 0; a < 3; a++) xw[3 * nc + a] = (float)(Tprev[a] * Xc[0] + Tprev[4 + a] * Xc[1] + Tprev[8 + a] * Xc[2]);
                    obs[2 * nc] = c.x; obs[2 * nc + 1] = c.y;
                    const float sig2 = powf(1.2f, (float)c.octave) * powf(1.2f, (float)c.octave);
                    s2[nc] = sig2; is2[nc] = 1.0f / sig2;
                    nc++;
                }
                float T[16];
                memcpy(T, Tprev, sizeof T);
                const double tp0 = now_ms();
                const int ni = orbp_pose_optimization(nc, obs.data(), nullptr, is2.data(), xw.data(), fx, fy, cx, cy, 0.f, T, outl.data());
                const double tp1 = now_ms();
                if (ni < 0) { fprintf(stderr, "pose optimisation failed: %s\n", orbp_last_error()); return 2; }
                n_inl += ni;
                if (k >= 5) t_pose.push_back(tp1 - tp0);
                const double err = std::fabs(T[3] + k * base) + std::fabs(T[7]) + std::fabs(T[11]);
                max_err = std::max(max_err, err / base);
                // per axis, and at checkpoints: the chain is dead reckoning (every frame's MapPoints are rebuilt through the previous
                // ESTIMATE), so the error is a random walk; z is the weakly observed axis of this fronto-parallel scene
                max_axis[0] = std::max(max_axis[0], std::fabs(T[3] + k * base) / base); max_axis[1] = std::max(max_axis[1], std::fabs((double)T[7]) / base);
                max_axis[2] = std::max(max_axis[2], std::fabs((double)T[11]) / base);
                if (k % 8 == 0 && n_ck < 8) { ck_frame[n_ck] = k; ck_err[n_ck] = err / base; n_ck++; }
                if (k % 8 == 0) {            // relocalisation: no prior, EPnP RANSAC then pose optimisation on its inliers' pose
                    const double tr0 = now_ms();
                    orbp_pnp *ps = nullptr;
                    CHK(orbp_pnp_create(&ps, nc, obs.data(), s2.data(), xw.data(), fx, fy, cx, cy));
                    CHK(orbp_pnp_set_ransac_parameters(ps, 0.99, 10, 300, 4, 0.5f, 5.991f));
                    float Tr[16]; int no_more = 0, ninl = 0, got = 0;
                    while (!got && !no_more) got = orbp_pnp_iterate(ps, 5, &no_more, inl.data(), &ninl, Tr);
                    orbp_pnp_destroy(ps);
                    if (got > 0) {
                        orbp_pose_optimization(nc, obs.data(), nullptr, is2.data(), xw.data(), fx, fy, cx, cy, 0.f, Tr, outl.data());
                        const double e = (std::fabs(Tr[3] + k * base) + std::fabs(Tr[7]) + std::fabs(Tr[11])) / base;
                        max_reloc_err = std::max(max_reloc_err, e);
                        n_reloc_ok++;
                        // src/Tracking.cc:1455-1462: with few inliers the reference projects the candidate key frame's other MapPoints
                        // through the PnP pose, ORBmatcher(0.9, true).SearchByProjection(mCurrentFrame, pKF, sFound, 10, 100), and
                        // optimises again.  Here on every relocalisation (to time and count it): the previous frame is the key
                        // frame, all its keypoints carry a MapPoint, sFound = the BoW matches PnP worked on.
                        {
                            const int np = prev.n;
                            std::vector<float> xk((size_t)np * 3), mfmax(np), pu(np), pv(np), d3(np), ang(np);
                            std::vector<uint8_t> inside(np), use(np, 1), has(cur.n, 0);
                            std::vector<int32_t> lv(np, 0), cmk(cur.n, -1);
                            for (int i = 0; i < np; i++) {
                                const orbx_keypoint &p = prev.kps[i];
                                const int px = std::min(std::max((int)lrintf(p.x), 0), W - 1), py = std::min(std::max((int)lrintf(p.y), 0), H - 1);
                                const double Z = depth[layer[(size_t)py * W + px]];
                                const double Xc[3] = {(p.x - cx) * Z / fx - Tprev[3], (p.y - cy) * Z / fy - Tprev[7], Z - Tprev[11]};
                                for (int a = 0; a < 3; a++) xk[3 * (size_t)i + a] = (float)(Tprev[a] * Xc[0] + Tprev[4 + a] * Xc[1] + Tprev[8 + a] * Xc[2]);
                                const double Pc[3] = {(p.x - cx) * Z / fx, (p.y - cy) * Z / fy, Z};                               // in the key frame's camera
                                mfmax[i] = (float)std::sqrt(Pc[0] * Pc[0] + Pc[1] * Pc[1] + Pc[2] * Pc[2]) * sfac[p.octave];   // MapPoint::UpdateNormalAndDepth
                                ang[i] = p.angle;
                            }
                            for (int i = 0; i < cur.n; i++)
                                if (match_f[i] >= 0) { has[i] = 1; use[match_f[i]] = 0; }                                          // sFound
                            const float bnd[4] = {0.f, (float)W, 0.f, (float)H};
                            CHK(orbm_project_points(Tr, fx, fy, cx, cy, bnd, xk.data(), np, pu.data(), pv.data(), nullptr, d3.data(), inside.data()));
                            for (int i = 0; i < np; i++) {
                                if (!use[i]) continue;
                                if (!inside[i] || d3[i] < 0.8f * (mfmax[i] / sfac[7]) || d3[i] > 1.2f * mfmax[i]) { use[i] = 0; continue; }
                                lv[i] = orbm_predict_scale(mfmax[i], d3[i], logf(1.2f), 8);
                            }
                            int nadd = 0;
                            CHK(orbm_search_by_projection_kf(mt, np, use.data(), pu.data(), pv.data(), lv.data(), prev.desc.data(), ang.data(), sfac, 8,
                                                             cur.kps.data(), cur.desc.data(), cur.n, 10.0f, 100, 1, has.data(), cmk.data(), &nadd));
                            n_reloc_add += nadd;
                        }
                    }
                    n_reloc++;
                    t_reloc.push_back(now_ms() - tr0);
                }
                memcpy(Tlast, Tprev, sizeof Tlast);
                memcpy(Tprev, T, sizeof T);
            }
            const double t4b = now_ms();
            if (pose) {
                // TrackWithMotionModel's matcher proper (src/Tracking.cc:879-883): every keypoint of the last frame carries a
                // MapPoint (its layer's depth through the last pose), the current pose stands in for the velocity model's
                // prediction, ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, LastFrame, 15, mono)
                for (int i = 0; i < prev.n; i++) {
                    const orbx_keypoint &p = prev.kps[i];
                    const int px = std::min(std::max((int)lrintf(p.x), 0), W - 1), py = std::min(std::max((int)lrintf(p.y), 0), H - 1);
                    const double Z = depth[layer[(size_t)py * W + px]];
                    const double Xc[3] = {(p.x - cx) * Z / fx - Tlast[3], (p.y - cy) * Z / fy - Tlast[7], Z - Tlast[11]};
                    for (int a = 0; a < 3; a++) xw_all[3 * i + a] = (float)(Tlast[a] * Xc[0] + Tlast[4 + a] * Xc[1] + Tlast[8 + a] * Xc[2]);
                }
                std::fill(cur_obs.begin(), cur_obs.end(), -1);
                const float bounds[4] = {0.f, (float)W, 0.f, (float)H};
                int nmp = 0;
                CHK(orbm_search_by_projection_last(mt, prev.n, ones.data(), xw_all.data(), prev.desc.data(), one_obs.data(), prev.kps.data(), Tprev, Tlast,
                                                   fx, fy, cx, cy, 0.f, 0.f, bounds, sfac, 8, cur.kps.data(), cur.desc.data(), nullptr, cur.n,
                                                   15.0f, 1, 1, cur_obs.data(), cur_match.data(), &nmp));
                nm_proj += nmp;
                if (dump.f) { dump.put(Tprev, 16); dump.put(Tlast, 16); dump.put(xw_all.data(), (size_t)3 * prev.n); dump.put(cur_match.data(), (size_t)cur.n); dump.i32(nmp); }
            } else {
            for (int i = 0; i < prev.n; i++) {           // motion-model windows around the previous positions
                const orbx_keypoint &p = prev.kps[i];
                qx[i] = p.x; qy[i] = p.y; qr[i] = 15.0f * powf(1.2f, (float)p.octave);
                mn[i] = std::max(p.octave - 1, -1); mx[i] = p.octave + 1; aq[i] = p.angle;
            }
            CHK(orbm_search_area_best2(mt, prev.desc.data(), qx.data(), qy.data(), qr.data(), mn.data(), mx.data(), prev.n,
                                       cur.desc.data(), nullptr, bi.data(), bd.data(), sd.data()));
            for (int i = 0; i < prev.n; i++) m12[i] = bd[i] <= ORBM_TH_HIGH ? bi[i] : -1;
            for (int i = 0; i < cur.n; i++) at[i] = cur.kps[i].angle;
            const int nrot = orbm_rot_filter(aq.data(), at.data(), m12.data(), prev.n);
            nm_proj += nrot;
            if (dump.f) { dump.put(qr.data(), (size_t)prev.n); dump.put(m12.data(), (size_t)prev.n); dump.i32(nrot); }
            }
            dump.close();
            const double t5 = now_ms();
            if (k >= 5) {
                const double d[5] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4b};
                for (int i = 0; i < 5; i++) t_stage[i].push_back(d[i]);
                t_all.push_back(t5 - t0);
            }
        }
        std::swap(cur, prev);
    }
    std::sort(t_all.begin(), t_all.end());
    const double med = t_all.empty() ? 0 : t_all[t_all.size() / 2];
    double sm[5];
    for (int i = 0; i < 5; i++) { std::sort(t_stage[i].begin(), t_stage[i].end()); sm[i] = t_stage[i].empty() ? 0 : t_stage[i][t_stage[i].size() / 2]; }
    if (pose) {
        std::sort(t_pose.begin(), t_pose.end()); std::sort(t_reloc.begin(), t_reloc.end());
        printf("{\"pose\": {\"ms_median_pose_optimization\": %.3f, \"ms_median_relocalization\": %.3f, \"inliers_per_frame\": %.1f, "
               "\"max_translation_error_in_baselines\": %.4f, \"relocalizations\": \"%ld/%ld\", \"max_reloc_error_in_baselines\": %.4f, "
               "\"reloc_search_by_projection_kf_matches\": %.1f, \"max_error_per_axis_in_baselines\": [%.4f, %.4f, %.4f], \"error_at_frame\": {",
               t_pose.empty() ? 0 : t_pose[t_pose.size() / 2], t_reloc.empty() ? 0 : t_reloc[t_reloc.size() / 2],
               (double)n_inl / std::max(K - 1, 1), max_err, n_reloc_ok, n_reloc, max_reloc_err, (double)n_reloc_add / std::max(n_reloc_ok, 1L),
               max_axis[0], max_axis[1], max_axis[2]);
        for (int i = 0; i < n_ck; i++) printf("%s\"%d\": %.4f", i ? ", " : "", ck_frame[i], ck_err[i]);
        printf("}}}\n");
    }
    printf("{\"harness\": \"C++ through the C ABI\", \"shape\": \"%dx%d n=%d\", \"frames_timed\": %d, \"ms_per_frame_median\": %.3f, \"frames_per_s\": %.1f, "
           "\"ms_median\": {\"extract\": %.3f, \"grid\": %.3f, \"bow\": %.3f, \"search_by_bow\": %.3f, \"search_by_projection\": %.3f}, "
           "\"matches_per_frame\": {\"bow\": %.1f, \"projection\": %.1f}}\n",
           W, H, NF, (int)t_all.size(), med, med > 0 ? 1000.0 / med : 0.0, sm[0], sm[1], sm[2], sm[3], sm[4],
           (double)nm_bow / std::max(K - 1, 1), (double)nm_proj / std::max(K - 1, 1));
    orbv_destroy(voc); orbm_destroy(mt); orbx_destroy(exs[0]); orbx_destroy(exs[1]);
    return 0;
}
