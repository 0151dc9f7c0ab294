/* mrt_debug.h — diagnostics, device-function probes and library-internal A/B switches of libmrt_hip.so.
 *
 * NOT part of the host contract (include/mrt_abi.h, which a reference-side binding uses — INTEGRATION.md) and not installed with it:
 * these entry points serve tests/, tools/ and bench.py; they come and go with the experiments that need them.  Plain C, like the ABI.  */
#ifndef MRT_DEBUG_H
#define MRT_DEBUG_H
#include "mrt_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- library-internal A/B switches (tests, tools/, bench.py --opt)
 * Not part of the host contract: keys come and go with the experiments that need them; every setting renders the same image bit for bit.
 * Today: "persistent" / "persist_chunk" / "wave_slots" / "stream_even" / "xcd_counters" (how a traversal launch splits its rays over waves), "primary_hint",
 * "primary_wide", "fuse_primary", "wide_bounce", "throughput_chain", "shadow_planes", "tail_accumulate", "tl_pairs" / "tl_pair_cap" (DESIGN.md §6).  Public keys are accepted too. */
int mrt_debug_renderer_set_option(MRTRenderer r, const char *key, double value);
int mrt_debug_renderer_get_option(MRTRenderer r, const char *key, double *value);

/* ---------------------------------------------------------------- device-function probes
 * Evaluate the kernel's helper functions on the device for known-answer tests
 * (Raytracing.metal:41-56 halton, :78-88 hemisphere, :132-147 align).                         */
int mrt_debug_halton(MRTContext ctx, const int32_t *i, const int32_t *d, size_t n, float *out);
int mrt_debug_hemisphere(MRTContext ctx, const float *u2, const float *normal3, size_t n, float *out3);
int mrt_debug_seeds(MRTContext ctx, uint32_t seed, int32_t width, int32_t height, uint32_t *out);
/* Diagnostics: per ray {node visits, leaf visits, triangle tests, hit gid, start tick, end tick, 0, 0}
 * (8 x uint32 each; ticks of the 100 MHz wall clock).                                           */
int mrt_debug_traversal_stats(MRTScene scene, const MRTRay *rays, size_t n, int32_t any_hit, uint32_t *out8);
/* Diagnostics: lane accounting of the wide stream traversal, per wave of `per_wave` rays:
 * {iterations, sum live lanes, sum node lanes, sum triangle lanes, refills, refilled lanes, hits, rays}.       */
int mrt_debug_stream_stats(MRTScene scene, const MRTRay *rays, size_t n, int32_t any_hit, uint32_t per_wave, uint32_t *out8, size_t nwaves);
/* The render kernels' own traversal (8-wide layout, wave-level stream with lane refill; TLAS + BLASes of an instanced scene in one loop)
 * on caller rays, for parity tests of that path against mrt_scene_intersect_closest / _any.  min_distance must be 0.  any_hit != 0:
 * only out[i].type (1 = occluded) is meaningful.                                                                                   */
int mrt_debug_intersect_stream(MRTScene scene, const MRTRay *rays, size_t n, int32_t any_hit, MRTIntersection *out);
/* Diagnostics: fill of the 8-wide nodes: out12[c] = nodes with c children (c = 0..8), [9] internal children, [10] leaf children, [11] triangles. */
int mrt_debug_wide_histogram(MRTScene scene, uint32_t *out12);
/* Diagnostics: the 8-wide nodes of a committed scene as they lie in device memory (80 bytes each); out == NULL: only the count.                      */
int mrt_debug_read_wnodes(MRTScene scene, void *out, size_t nbytes, uint64_t *num_nodes);
/* Diagnostics: commits of this scene served by a refit (mrt_scene_update_mesh) since its last build.   */
int mrt_debug_scene_refits(MRTScene scene, uint32_t *out);
/* Diagnostics: host wall time (ms) of the last mrt_scene_commit of a flattened scene by phase: {upload staging, device allocations + upload enqueue,
 * topology (flatten .. refit, with its read-backs), 8-wide emit, rope emit, validation}.                                                              */
int mrt_debug_commit_times(MRTScene scene, double *out6);
/* The index validation mrt_scene_commit runs (scene option "validate", default 1): every child / packet / instance index of the committed
 * 8-wide layout, the instance rows and the TLAS lies inside its array and children come after their parents; MRT_ERR_STATE + message
 * otherwise.  mrt_debug_validate_patched runs it on a COPY of the 8-wide nodes in which one 32-bit word (0..19) of one node is replaced —
 * the validator's own test; the scene's arrays are not touched.  (A diagnostics build, -DMRT_DIAGNOSTICS, also has mrt_debug_poke_wnode,
 * which overwrites the word in the live array; the release library does not export it.)                                                */
int mrt_debug_validate(MRTScene scene);
int mrt_debug_validate_patched(MRTScene scene, uint32_t node, uint32_t word, uint32_t value);
#ifdef MRT_DIAGNOSTICS
int mrt_debug_poke_wnode(MRTScene scene, uint32_t node, uint32_t word, uint32_t value, uint32_t *old_value);
#endif
/* Scene option "builder" = 2's host part on caller boxes (n x float4 lo, n x float4 hi): a top-down binned-SAH binary tree — leaf order[n], left /
 * right of the n - 1 internal nodes (ids 0 .. n-2; leaf at position j = id n-1+j), parent of all 2n - 1 nodes (0xFFFFFFFF = root).  No device needed. */
int mrt_debug_host_sah(const float *lo4, const float *hi4, uint32_t n, uint32_t *order, uint32_t *left, uint32_t *right, uint32_t *parent);
/* The size check mrt_scene_commit applies (host only, no device needed): MRT_OK, or MRT_ERR_UNSUPPORTED when a scene of
 * `triangles` triangles whose BVH keeps `nodes` nodes (0 = unknown) cannot be addressed by the traversal layouts.      */
int mrt_debug_layout_limits(uint64_t triangles, uint64_t nodes);
/* Calibration of the ceilings the render kernels are priced against (bench.py): out5 = {wave64 v_fma_f32 instructions/s
 * with every SIMD holding 8 waves, the same for v_pk_fma_f32, bytes/s of divergent 16-byte gathers from a table of about
 * table_bytes, bytes/s of divergent 80-byte records (the wide-node fetch) from such a table, the shader clock in Hz
 * observed during the v_fma_f32 loop}.                                                                                 */
int mrt_debug_calibrate(MRTContext ctx, size_t table_bytes, double *out5);

#ifdef __cplusplus
}
#endif
#endif /* MRT_DEBUG_H */
