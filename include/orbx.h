/*
 * orbx.h -- C ABI of the MI355X-native ORB extractor (liborbx.so).
 *
 * Drop-in boundary for the reference's ORB_SLAM2::ORBextractor (citations relative to the
 * reference tree, WChen09/My-SLAM):
 *   include/ORBextractor.h:51-52   ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 *   include/ORBextractor.h:59-61   operator()(image, mask, keypoints, descriptors)
 *   include/ORBextractor.h:63-83   GetLevels / GetScaleFactor / GetScaleFactors / ...
 *   include/ORBextractor.h:85      public mvImagePyramid
 * The C++ adapter my-slam_amd/host/ORBextractor.h keeps those exact names on top of this ABI;
 * INTEGRATION.md shows the binding a maintainer adds to Frame.cc / Tracking.cc.
 *
 * Plain C types only: pointers, sizes, ints.  The library never allocates caller-visible memory and
 * never throws; every entry point returns an orbx_status (0 = ok, < 0 = error, text via
 * orbx_last_error()).  One extractor handle owns one HIP stream and its workspace; handles are
 * independent and may be used from different threads concurrently (the reference runs two
 * extractors on two threads for stereo, src/Frame.cc:78-81).  A single handle is not re-entrant,
 * like the reference object (it overwrites mvImagePyramid every call, src/ORBextractor.cc:1116).
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_MAX_LEVELS 16

typedef enum {
    ORBX_OK = 0,
    ORBX_E_INVALID = -1,        /* bad argument */
    ORBX_E_CAPACITY = -2,       /* caller buffer smaller than orbx_capacity() / result count */
    ORBX_E_SHAPE = -3,          /* image shape the reference itself cannot process (UB there) or > max */
    ORBX_E_HIP = -4,            /* HIP runtime error (no GPU, OOM, launch failure) */
    ORBX_E_CAND_OVERFLOW = -5,  /* more FAST candidates than the per-level candidate buffer */
    ORBX_E_TREE_OVERFLOW = -6   /* quadtree node arena exhausted */
} orbx_status;

/* Same field layout as cv::KeyPoint (28 bytes): Point2f pt; float size, angle, response; int octave, class_id */
typedef struct {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} orbx_keypoint;

typedef struct orbx_extractor orbx_extractor;

enum {
    ORBX_OPT_BLUR_ROUNDING = 1, /* 0: OpenCV portable C column pass (default); 1: OpenCV x86 SSE2 column pass */
    ORBX_OPT_OVERLAP_PYRAMID = 3, /* 0/1 (default 0; measured slower on MI355X): resize chain on a side stream underneath FAST on level 0 */
    ORBX_OPT_SUBBATCHES = 2,    /* 1..4 (default 1): batches of >= 16 frames are cut into this many sub-batches on
                                   separate HIP streams so latency-bound and VALU-bound kernels overlap */
    ORBX_OPT_BATCH_CHUNK = 4    /* frames per chunk of orbx_extract_batch's upload / extract / download pipeline (default 16;
                                   0 = the whole batch in one piece) */
};

/*
 * ORBextractor::ORBextractor (include/ORBextractor.h:51).  `device` is the HIP device ordinal;
 * max_width/max_height/max_batch size the workspace (pyramids, candidate and tree buffers) once.
 */
int orbx_create(orbx_extractor **out, int nfeatures, float scale_factor, int nlevels,
                int ini_th_fast, int min_th_fast, int device,
                int max_width, int max_height, int max_batch);
void orbx_destroy(orbx_extractor *h);
int orbx_set_option(orbx_extractor *h, int option, int value);

/* include/ORBextractor.h:63-83 getters.  Arrays must hold nlevels floats. */
int orbx_get_levels(const orbx_extractor *h);
float orbx_get_scale_factor(const orbx_extractor *h);
int orbx_get_tables(const orbx_extractor *h, float *scale_factors, float *inv_scale_factors,
                    float *level_sigma2, float *inv_level_sigma2);
int orbx_get_features_per_level(const orbx_extractor *h, int *quota);
/* keypoints one frame can produce at most (the reference does not cap at nfeatures) */
int orbx_capacity(const orbx_extractor *h);

/*
 * The same call in two halves, for callers that have host work to do while the GPU extracts (e.g. the pose solvers of
 * frame k while frame k+1 is in flight on a second handle): orbx_extract_begin stages the image and enqueues copies and
 * kernels on the handle's stream without waiting; orbx_extract_end waits and fills the outputs.  One call may be in flight
 * per handle; `image` may be released as soon as begin returns.  begin on an empty image followed by end gives *n = 0.
 * From the second call of a shape on, begin replays one HIP graph (upload, kernels, download) instead of a dozen launches.
 */
int orbx_extract_begin(orbx_extractor *h, const uint8_t *image, int width, int height, int stride);
int orbx_extract_end(orbx_extractor *h, orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n);

/*
 * ORBextractor::operator() (include/ORBextractor.h:59).  Host image (CV_8UC1, `stride` bytes per
 * row), host outputs with room for `cap` >= orbx_capacity() entries; *n receives the count.
 * An empty image (NULL / w<=0 / h<=0) is the reference's silent return: ORBX_OK with *n = 0.
 * Keypoint order and descriptor rows follow the reference: level 0..L-1, quadtree list order inside.
 */
int orbx_extract(orbx_extractor *h, const uint8_t *image, int width, int height, int stride,
                 orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n);

/* Same over `nframes` equally sized host frames (frame k at images + k*frame_stride). Outputs are
 * [nframes][cap] keypoints, [nframes][cap][32] descriptor bytes, counts[nframes]. */
int orbx_extract_batch(orbx_extractor *h, const uint8_t *images, int nframes, int width, int height,
                       int row_stride, size_t frame_stride,
                       orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *counts);

/*
 * The batched-frames mode sharded over several handles, normally one per GPU of the node (BASELINE north_star):
 * one process, one host thread per handle.  Handle i takes the i-th contiguous block of the batch (sizes differ by at most one)
 * and needs max_batch >= that block; outputs as for orbx_extract_batch, at the frames' own positions, so the result does not
 * depend on the split.  Frames are independent: there is no collective on this path.
 */
int orbx_extract_batch_multi(orbx_extractor *const *handles, int nhandles, const uint8_t *images, int nframes, int width,
                             int height, int row_stride, size_t frame_stride, orbx_keypoint *keypoints,
                             uint8_t *descriptors, int cap, int *counts);

/*
 * Device-resident batch: all pointers are HIP device pointers on the handle's device; the work is
 * enqueued on `hip_stream` (a hipStream_t, NULL = the handle's own stream) and NOT synchronised.
 * d_status[nframes] receives an orbx_status per frame.  The input must stay valid until the stream
 * has drained (level 0 of the pyramid is the input itself).
 */
int orbx_extract_batch_device(orbx_extractor *h, const uint8_t *d_images, int nframes, int width,
                              int height, int row_stride, size_t frame_stride,
                              orbx_keypoint *d_keypoints, uint8_t *d_descriptors, int cap,
                              int32_t *d_counts, int32_t *d_status, void *hip_stream);

/* mvImagePyramid access (include/ORBextractor.h:85): size of level `level` for the last extracted
 * shape, and a copy of that level of frame `frame` of the last call into host memory.
 * border = 0 copies the w x h interior; border = 19 reproduces the reference's
 * (w+38) x (h+38) reflect-101 bordered buffer (src/ORBextractor.cc:1115-1133). */
int orbx_level_size(const orbx_extractor *h, int level, int *width, int *height);
int orbx_download_level(orbx_extractor *h, int frame, int level, uint8_t *dst, int dst_stride, int border);
/* All levels at once (one synchronisation): dst[l] / dst_stride[l] per level, sized (w_l + 2*border) x (h_l + 2*border). */
int orbx_download_pyramid(orbx_extractor *h, int frame, uint8_t *const *dst, const int *dst_stride, int border);

/* Debug/parity taps for the last call (host copies): FAST candidates of one level before the
 * quadtree, as (x, y, response) int32 triples in unspecified order.  Returns the count or < 0. */
int orbx_download_candidates(orbx_extractor *h, int frame, int level, int32_t *xyr, int cap);

/* Elapsed GPU milliseconds of the last call per stage, measured with HIP events on the stream the
 * kernels ran on: [0] pyramid, [1] fast, [2] quadtree, [3] describe (orientation+blur+rBRIEF). */
int orbx_last_stage_ms(orbx_extractor *h, float ms[4]);
/* mode 0 = off; 1 = time every call in isolation (the call synchronises on its last event; read with orbx_last_stage_ms);
 * 2 = drop the stage events of the last ORBX_PROF_RING calls into the stream and never wait for them: for timing kernels
 * while several handles run on several streams.  Read with orbx_stage_ms_ring after synchronising the stream yourself. */
#define ORBX_PROF_RING 16
int orbx_set_profiling(orbx_extractor *h, int mode);
/* ms[4 * i + k]: stage k of the i-th newest profiled call (mode 2).  Returns the number of calls written (<= max_calls). */
int orbx_stage_ms_ring(orbx_extractor *h, float *ms, int max_calls);

/*
 * "next" row N3: Frame::ComputeStereoMatches (src/Frame.cc:466-640) on the GPU.  `left` and
 * `right` are the two extractor handles of the stereo rig (src/Frame.cc:78-81); the pyramids of their last
 * orbx_extract call (mvImagePyramid) are read in place.  Inputs are that call's keypoints/descriptors
 * (host); mb = mbf/fx, mbf as in Frame.  Outputs mvuRight / mvDepth (nl floats, -1 = no match).
 */
int orbx_stereo_matches(orbx_extractor *left, orbx_extractor *right,
                        const orbx_keypoint *kl, const uint8_t *dl, int nl,
                        const orbx_keypoint *kr, const uint8_t *dr, int nr,
                        float mb, float mbf, float *u_right, float *depth);

const char *orbx_last_error(void);
const char *orbx_version(void);

#ifdef __cplusplus
}
#endif
#endif
