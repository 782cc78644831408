// Exercises the C++ adapters (my-slam_amd/host) the way src/Frame.cc:247-253 uses the reference
// class: ORBextractor(...)(im, cv::Mat(), keys, descriptors), with the reference's five-argument constructor
// (src/Tracking.cc:121).  Prints a checksum the pytest wrapper compares with the Python/C-ABI result.
// With argv[1] == "compile-only" nothing runs.  (The Tracking-thread matcher call sites: tracking_callsites.cc.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "PnPsolver.h"

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "compile-only")) return 0;
    if (argc > 1 && !strcmp(argv[1], "pose")) {
        // host-only: a camera at t = (0.3, -0.1, 0.2) looking at 60 points, 20 of them mismatched; Relocalization's call
        // sequence (src/Tracking.cc:1392-1445): PnPsolver -> SetRansacParameters -> iterate(5) -> PoseOptimization
        const float fx = 718.856f, fy = 718.856f, cx = 607.19f, cy = 185.2f, t[3] = {0.3f, -0.1f, 0.2f};
        std::vector<float> p2d, s2, p3d, is2;
        std::vector<size_t> idx;
        unsigned long long r = 12345;
        auto u01 = [&]() { r = r * 6364136223846793005ull + 1442695040888963407ull; return (double)(r >> 11) / 9007199254740992.0; };
        for (int i = 0; i < 60; i++) {
            const float X = (float)(u01() * 8 - 4), Y = (float)(u01() * 4 - 2), Z = (float)(4 + 16 * u01());
            float u = fx * (X + t[0]) / (Z + t[2]) + cx, v = fy * (Y + t[1]) / (Z + t[2]) + cy;
            if (i % 3 == 0) { u += 40; v -= 35; }
            p3d.insert(p3d.end(), {X, Y, Z}); p2d.insert(p2d.end(), {u, v}); s2.push_back(1.f); is2.push_back(1.f); idx.push_back(2 * i);
        }
        srand(1);
        ORB_SLAM2::PnPsolver solver(p2d, s2, p3d, idx, 120, fx, fy, cx, cy);
        solver.SetRansacParameters(0.99, 10, 300, 4, 0.5, 5.991);
        std::vector<bool> inl; int nInl = 0; bool noMore = false;
        std::optional<ORB_SLAM2::Pose> T;
        while (!T && !noMore) T = solver.iterate(5, noMore, inl, nInl);
        if (!T) { printf("no pose\n"); return 1; }
        std::vector<bool> outl;
        ORB_SLAM2::Pose Tcw = *T;
        const int good = ORB_SLAM2::Optimizer::PoseOptimization(p2d, {}, is2, p3d, fx, fy, cx, cy, 0.f, Tcw, outl);
        int wrongKept = 0;
        for (int i = 0; i < 60; i += 3) wrongKept += !outl[i];
        printf("%d %d %d %.4f %.4f %.4f\n", nInl, good, wrongKept, Tcw[3], Tcw[7], Tcw[11]);
        return 0;
    }
    if (argc < 4) { fprintf(stderr, "usage: %s raw.u8 W H\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    cv::Mat im(H, W, CV_8UC1);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(im.data, 1, (size_t)W * H, f) != (size_t)W * H) { fprintf(stderr, "read failed\n"); return 2; }
    fclose(f);
    ORB_SLAM2::ORBextractor extractor(1000, 1.2f, 8, 20, 7);
    if (!extractor.Valid()) { fprintf(stderr, "create failed: %s\n", extractor.LastError().c_str()); return 3; }
    std::vector<cv::KeyPoint> keys;
    cv::Mat descriptors;
    extractor(im, cv::Mat(), keys, descriptors);
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < keys.size(); i++) {
        const unsigned char *p = (const unsigned char *)&keys[i];
        for (int b = 0; b < 28; b++) { h ^= p[b]; h *= 1099511628211ull; }
        const unsigned char *d = descriptors.ptr<unsigned char>((int)i);
        for (int b = 0; b < 32; b++) { h ^= d[b]; h *= 1099511628211ull; }
    }
    {   // the two-halves form gives the same frame
        std::vector<cv::KeyPoint> keys2;
        cv::Mat desc2;
        if (!extractor.Begin(im)) { fprintf(stderr, "Begin failed: %s\n", extractor.LastError().c_str()); return 3; }
        extractor.End(keys2, desc2);
        if (keys2.size() != keys.size() || memcmp(keys2.data(), keys.data(), keys.size() * sizeof(cv::KeyPoint)) ||
            memcmp(desc2.data, descriptors.data, keys.size() * 32)) { fprintf(stderr, "Begin/End differs from operator()\n"); return 4; }
    }
    ORB_SLAM2::ORBmatcher matcher(0.9f, true);
    std::vector<int32_t> bi, bd, sd;
    matcher.BestTwo(descriptors, descriptors, nullptr, nullptr, bi, bd, sd);
    int self = 0;
    for (size_t i = 0; i < bi.size(); i++) self += (bd[i] == 0);
    // a 3-channel / float image is not CV_8UC1: empty result (the reference asserts, src/ORBextractor.cc:1052)
    std::vector<cv::KeyPoint> keysF;
    cv::Mat descF, imF(H, W, CV_32FC1);
    extractor(imF, cv::Mat(), keysF, descF);
    const int rejected = keysF.empty() && descF.empty();
    // a larger image than the handle was sized for (1920 x 1080 at first): the handle grows, no failure
    cv::Mat big(1200, 2000, CV_8UC1);
    for (int y = 0; y < big.rows; y++) for (int x = 0; x < big.cols; x++) big.at<unsigned char>(y, x) = im.at<unsigned char>(y % H, x % W);
    std::vector<cv::KeyPoint> keysB;
    cv::Mat descB;
    extractor(big, cv::Mat(), keysB, descB);
    const int grew = keysB.size() > 900 && descB.rows == (int)keysB.size();
    extractor(im, cv::Mat(), keys, descriptors);
    // mvImagePyramid is valid after operator() (include/ORBextractor.h:85), as Frame::ComputeStereoMatches needs it (src/Frame.cc:473,563,580):
    // level 0's interior is the image, one step outside the interior is the reflect-101 border, every level hashes to what the C ABI returns
    int pyr_ok = (int)extractor.mvImagePyramid.size() == 8 && extractor.mvImagePyramid[0].rows == H && extractor.mvImagePyramid[0].cols == W;
    unsigned long long hp = 1469598103934665603ull;
    for (int l = 0; pyr_ok && l < 8; l++) {
        const cv::Mat &lv = extractor.mvImagePyramid[l];
        for (int y = 0; y < lv.rows; y++) {
            const unsigned char *row = lv.ptr<unsigned char>(y);
            for (int x = 0; x < lv.cols; x++) { hp ^= row[x]; hp *= 1099511628211ull; }
            pyr_ok = pyr_ok && row[-1] == row[1] && row[-19] == row[19] && row[lv.cols] == row[lv.cols - 2];
        }
        const unsigned char *r0 = lv.ptr<unsigned char>(0);
        pyr_ok = pyr_ok && !memcmp(r0 - (ptrdiff_t)lv.step - 19, lv.ptr<unsigned char>(1) - 19, (size_t)lv.cols + 38) &&
                 !memcmp(lv.ptr<unsigned char>(lv.rows - 1) + (ptrdiff_t)19 * (ptrdiff_t)lv.step - 19, lv.ptr<unsigned char>(lv.rows - 20) - 19, (size_t)lv.cols + 38);
    }
    for (int y = 0; pyr_ok && y < H; y++) pyr_ok = !memcmp(extractor.mvImagePyramid[0].ptr<unsigned char>(y), im.ptr<unsigned char>(y), (size_t)W);
    printf("%zu %016llx %d %d %d %d %d %d %016llx\n", keys.size(), h, self, extractor.mvImagePyramid[7].cols,
           ORB_SLAM2::ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(1)), rejected, grew, pyr_ok, hp);
    return 0;
}
