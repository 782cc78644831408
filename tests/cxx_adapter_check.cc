// Exercises the C++ adapters (my-slam_amd/host) the way src/Frame.cc:247-253 uses the reference
// class: ORBextractor(...)(im, cv::Mat(), keys, descriptors).  Prints a checksum the pytest wrapper
// compares with the Python/C-ABI result.  With argv[1] == "compile-only" nothing runs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../my-slam_amd/host/ORBextractor.h"
#include "../my-slam_amd/host/ORBmatcher.h"

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "compile-only")) return 0;
    if (argc < 4) { fprintf(stderr, "usage: %s raw.u8 W H\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    cv::Mat im(H, W, cv::CV_8U);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(im.data, 1, (size_t)W * H, f) != (size_t)W * H) { fprintf(stderr, "read failed\n"); return 2; }
    fclose(f);
    ORB_SLAM2::ORBextractor extractor(1000, 1.2f, 8, 20, 7, 0, W, H);
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
    ORB_SLAM2::ORBmatcher matcher(0.9f, true);
    std::vector<int32_t> bi, bd, sd;
    matcher.BestTwo(descriptors, descriptors, nullptr, nullptr, bi, bd, sd);
    int self = 0;
    for (size_t i = 0; i < bi.size(); i++) self += (bd[i] == 0);
    extractor.FetchImagePyramid();
    printf("%zu %016llx %d %d %d\n", keys.size(), h, self, extractor.mvImagePyramid[7].cols,
           ORB_SLAM2::ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(1)));
    return 0;
}
