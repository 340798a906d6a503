// rt_records.h -- the register-resident form of a small scene.
//
// A scene of at most 64 records is kept, for the whole life of a wave, in eight
// VGPRs: lane L holds the eight dwords of record L.  The wave-uniform scan then
// fetches record i with v_readlane (VGPR -> SGPR, no memory access, no latency to
// hide), and a lane fetches "its" record (the primitive it hit, its material, the
// light it sampled) with ds_bpermute.  The Cornell box needs 33 records.
//
// Record kinds and dword layout (d0..d6 are floats unless noted, d7 is the kind word):
//   INNER     d0-2 box min, d3 (int) stream index to continue at when the box is missed, d4-6 box max
//   RECT      d0 a0, d1 a1, d2 b0, d3 b1, d4 k, d5 world normal x, d6 world normal z
//             (world normal y is +-1 for an xz rect and 0 otherwise: chains are y-rotations and shifts)
//   SPHERE    d0-2 centre IN WORLD SPACE (a rigid chain is folded in), d3 radius, d4 1/radius
//   BOX       d0-2 p0, d4-6 p1 in the space of its chain (src/box.h: six rects, here ONE slab test); always followed by
//   BOX_AUX   d0-2 the box centre IN WORLD SPACE, d3/d4 x and z of the world image of the box's x axis (the image of
//             its z axis is (-d4, 0, d3): chains are y-rotations and shifts); read after the scan to name the face
//   XFORM_T   d0-2 offset                     (src/transforms.h:14-28)
//   XFORM_R   d0 sin_theta, d1 cos_theta      (src/transforms.h:102-130)
//   MATERIAL  d0 (int) rt_material_kind, d1 (int) texture index or -1 when the colour is inline,
//             d2 p0 (fuzz / ref_idx), d3-5 constant colour
//   LIGHT     d0-4 as RECT (xz) or d0-3 as SPHERE, d5 rect area
// kind word:  bits 0-3 kind | 4-5 rect axis (0 xy, 1 xz, 2 yz; 3 for BOX and BOX_AUX) | 6 flip | 7 first of its list | 8 last of its list
//             | 9-14 first record of the instance chain | 15-17 number of chain records
//             | 18-25 material record (relative to materials_base)
//             | 26-31 stream records: RANK = where the reference's depth-first walk meets the primitive (its last visit);
//               of two hits at exactly the same t the one met later wins (src/bvh.h:132-139), whatever the stream's order
#ifndef RT_RECORDS_H_
#define RT_RECORDS_H_

#include <stdint.h>

#include "../../include/rt_amd.h"

#define RT_REC_MAX 64
#define RT_REC_FLAT_LEAVES 16 /* scenes with at most this many primitive records are stored without inner nodes */

enum rt_rec_kind {
    RT_REC_INNER = 0,
    RT_REC_RECT = 1,
    RT_REC_SPHERE = 2,
    RT_REC_XFORM_T = 3,
    RT_REC_XFORM_R = 4,
    RT_REC_MATERIAL = 5,
    RT_REC_LIGHT_RECT = 6,
    RT_REC_LIGHT_SPHERE = 7,
    RT_REC_LIGHT_NONE = 8,
    RT_REC_BOX = 9,
    RT_REC_BOX_AUX = 10
};

#define RT_KW_AXIS_SHIFT 4
#define RT_KW_FLIP (1u << 6)
#define RT_KW_FIRST (1u << 7)
#define RT_KW_LAST (1u << 8)
#define RT_KW_CHAIN_SHIFT 9   /* 9 bits: first record (6) | count (3) << 6 */
#define RT_KW_CHAIN_MASK 0x1ffu
#define RT_KW_MAT_SHIFT 18
#define RT_KW_MAT_MASK 0xffu
#define RT_KW_RANK_SHIFT 26   /* 6 bits */
#define RT_KW_RANK_MASK 0x3fu
/* The scan's per-lane best hit is ONE 64-bit number, smaller = better: t's bits (t > 0, so they order like t) above a key
 * of the record.  The key is the record's kind word re-arranged: 63 - rank in the top six bits (so that, of two hits at
 * exactly the same t, the one the reference's walk meets later is the smaller number), the material, and the stream
 * index in place of the chain bits; kind, axis and flip stay where they are.  A hit replaces the best iff its number is
 * smaller, and afterwards the low word of the best IS what shading needs to know about the record: no fetch. */
#define RT_REC_KEY(kw, index) (((RT_KW_RANK_MASK - (((kw) >> RT_KW_RANK_SHIFT) & RT_KW_RANK_MASK)) << RT_KW_RANK_SHIFT) | \
                               ((kw) & ((RT_KW_MAT_MASK << RT_KW_MAT_SHIFT) | 0x1ffu)) | ((uint32_t)(index) << RT_KEY_INDEX_SHIFT))
#define RT_KEY_INDEX_SHIFT 9
#define RT_KEY_INDEX(key) (((key) >> RT_KEY_INDEX_SHIFT) & 63u)
#define RT_REC_KEY_NONE 0xffffffffu

typedef struct rt_record_table {
    uint32_t rec[RT_REC_MAX + 1][8]; /* one spare record: the scan prefetches record i+1 */
    int32_t user_id[RT_REC_MAX]; /* rt_prim.user_id of primitive records (for rt_trace_rays) */
    int32_t face_user[RT_REC_MAX][6]; /* BOX records: user_id of the face rects, [2*axis + side] (axis x,y,z; side 0 at p0, 1 at p1) */
    int32_t prim[RT_REC_MAX];    /* rt_prim index of primitive records (BOX: the list) */
    int32_t n_stream;            /* records [0, n_stream) are the traversal stream */
    int32_t materials_base;
    int32_t lights_base, n_lights;
    int32_t n_records;
    int32_t flat;                /* 1: the stream holds no inner (box) records */
    /* flat streams are stored in RUNS: for each instance chain (world space first) its rects by axis (xy, xz, yz) and its
     * boxes (two records each), then all spheres (world space).
     * run[r] = n_box << 24 | chain word << 15 | n_xy << 10 | n_xz << 5 | n_yz.  n_runs == 0: the stream has more than
     * RT_REC_MAX_RUNS chains (or a run above 31 rects of an axis / 15 boxes); the generic scan then decodes every record. */
    uint32_t run[4];
    int32_t n_runs, n_spheres;
} rt_record_table;
#define RT_REC_MAX_RUNS 4

#ifdef __cplusplus
extern "C" {
#endif
/* Builds the table from a validated description.  Returns 1 if the scene fits
 * (<= 64 records, only spheres / rects / boxes / flat lists, no media, no moving
 * spheres, no texture that needs (u,v)), 0 if it does not (the caller then uses
 * the generic memory-resident path).  `why` receives a short reason.  keep_boxes: small scenes keep their inner
 * (box) records too (RT_CREATE_KEEP_BOX_RECORDS). */
int rt_records_build(const rt_scene_desc *d, int keep_boxes, rt_record_table *out, const char **why);
/* Lights only (n_stream = 0): for scenes that do not fit.  Returns 0 if there are more than 64 lights. */
int rt_records_build_lights(const rt_scene_desc *d, rt_record_table *out);
#ifdef __cplusplus
}
#endif

#endif /* RT_RECORDS_H_ */
