/*
 * orbv.h -- C ABI of the vocabulary-tree (DBoW2) primitives of liborbx.so: "next" row N2 of SURVEY.md 8(f).
 *
 * Reference (WChen09/My-SLAM, Thirdparty/DBoW2/DBoW2):
 *   TemplatedVocabulary.h:1338-1424  loadFromTextFile   (header "k L scoring weighting", then one node per line:
 *                                                        parent isLeaf d0 .. d31 weight)
 *   TemplatedVocabulary.h:1218-1259  transform(feature, word_id, weight, nid, levelsup): greedy descent, Hamming
 *                                    distance (FORB.cpp:81-101) to the children of the current node, first child
 *                                    wins a tie; nid = the node at level L - levelsup
 *   TemplatedVocabulary.h:1127-1194  transform(features, BowVector, FeatureVector, levelsup)
 *   BowVector.cpp:34-84, FeatureVector.cpp:31-45, ScoringObject.cpp:23-68 (L1 score)
 * Caller: Frame::ComputeBoW (src/Frame.cc:395-402, levelsup = 4); consumers: ORBmatcher::SearchByBoW
 * (src/ORBmatcher.cc:159-288: merge-join of two FeatureVectors, then the best/second-best loop = orbm_best2 /
 * orbm_distances on the CSR lists of orbv_feature_vector) and KeyFrameDatabase (score).
 *
 * The tree descent runs on the GPU (one wave per descriptor, one child per lane); the two tiny ordered maps are
 * assembled on the host with DBoW2's contents and order (features sorted by (id, feature index), the same double
 * additions in the same order).  Deviation where the reference is
 * undefined: a trailing empty line of the text file makes loadFromTextFile append a node with an uninitialised
 * descriptor under the root; this loader ignores empty lines.
 */
#ifndef ORBV_H
#define ORBV_H

#include <stddef.h>
#include <stdint.h>
#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORBV_TF_IDF = 0, ORBV_TF = 1, ORBV_IDF = 2, ORBV_BINARY = 3 };                 /* DBoW2::WeightingType */
enum { ORBV_L1_NORM = 0, ORBV_L2_NORM = 1, ORBV_CHI_SQUARE = 2, ORBV_KL = 3, ORBV_BHATTACHARYYA = 4, ORBV_DOT_PRODUCT = 5 };

typedef struct orbv_vocabulary orbv_vocabulary;

int orbv_load_text(orbv_vocabulary **out, const char *path, int device);
void orbv_destroy(orbv_vocabulary *v);
int orbv_info(const orbv_vocabulary *v, int *k, int *L, int *nnodes, int *nwords, int *scoring, int *weighting);

/* Per-feature part of transform(): word id, node id at level L - levelsup, word weight (0 = stopped word). */
int orbv_transform_features(orbv_vocabulary *v, const uint8_t *desc, int n, int levelsup,
                            int32_t *word_id, int32_t *node_id, double *weight);

/* BowVector of those features (ascending word ids, normalised as the scoring type demands).  Returns the count. */
int orbv_bow_vector(const orbv_vocabulary *v, const int32_t *word_id, const double *weight, int n,
                    int32_t *ids, double *vals, int cap);
/* FeatureVector: ascending node ids; node i holds features idx[off[i] .. off[i+1]) in feature order. */
int orbv_feature_vector(const int32_t *node_id, const double *weight, int n,
                        int32_t *node_ids, int32_t *off, int32_t *idx, int cap_nodes);
/* L1Scoring::score of two BowVectors */
double orbv_score_l1(const int32_t *ids1, const double *vals1, int n1, const int32_t *ids2, const double *vals2, int n2);

const char *orbv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
