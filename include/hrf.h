/*
 * include/hrf.h -- C ABI of libhrf_hip.so, the MI355X (gfx950) implementation of HumanRF's
 * ray-marching hot path. Plain pointers and sizes only: no torch / pybind types cross this line.
 *
 * Conventions (all entry points):
 *   - return 0 on success; non-zero on error, message via hrf_last_error() (thread-local).
 *     Mirrors the reference's std::runtime_error -> RuntimeError convention
 *     (actorshq/toolbox/native/utils.cuh:5-19, actorshq/dataset/native/occupancy_grid.cu:60-63).
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - the library never allocates or frees caller memory (ownership as in the reference:
 *     ray_sampler.cu:233-235, tensor_composition.cu:140,184-188 allocate through torch, i.e. the caller);
 *     the only library-owned memory is the occupancy ring (hrf_occgrid_*), RAII like
 *     occupancy_grid.cu:45-55.
 *   - every launch goes to the hipStream_t passed in `stream` (the reference launches on the legacy
 *     default stream, ray_sampler.cu:240,299; passing the caller's stream makes the path re-entrant).
 *   - no entry point synchronises the device or the stream.
 *
 * Reference interfaces replaced (file:line under /root/reference):
 *   hrf_occgrid_*            actorshq/dataset/native/occupancy_grid.cu:8-95      (class OccupanyGrid)
 *   hrf_sampler_*            actorshq/dataset/native/ray_sampler.cu:196-333      (get_{rays,samples}_{aabb,occupancy}_minmax)
 *   hrf_pool_replace         actorshq/dataset/data_loader.py:396-511             (replacer thread: refill of pool slots)
 *   hrf_compose_*            humanrf/scene_representation/native/tensor_composition.cu:120-225
 *   hrf_query_prep           humanrf/volume_rendering.py:63-72,109-119 + humanrf/scene_representation/humanrf.py:159-177
 *   hrf_encode4d_*           humanrf/scene_representation/decomposition4d.py:124-135 (4x tcnn HashGrid + compose)
 *   hrf_hashgrid_*           humanrf/scene_representation/decomposition4d.py:79-122  (ONE tcnn.Encoding, stand-alone)
 *   hrf_density_mlp_fwd      humanrf/scene_representation/humanrf.py:181-186     (tcnn FullyFusedMLP + truncated_exp)
 *   hrf_color_mlp_fwd        humanrf/scene_representation/humanrf.py:188-208     (tcnn Composite encoding + FullyFusedMLP)
 *   hrf_mlp_bwd              autograd of the two above (tcnn backward + humanrf/utils/activation.py:23-29)
 *   hrf_density_mlp_bwd, hrf_color_mlp_bwd  the same, one network at a time (tcnn.Network / NetworkWithInputEncoding backward)
 *   hrf_visibility           humanrf/volume_rendering.py:75-84                   (nerfacc.render_visibility + compaction)
 *   hrf_prune_march/pack     humanrf/volume_rendering.py:42-84                   (whole prune_samples body, fused, early termination)
 *   hrf_composite_*          humanrf/volume_rendering.py:123-145                 (nerfacc weights + accumulate + bg blend)
 *   hrf_weights_*, hrf_accumulate_*  the same nerfacc 0.3.1 calls (volume_rendering.py:123-141) as stand-alone ops
 *   hrf_ray_segment_order*   (no counterpart: schedule of the march over the 8 XCDs of the MI355X; frame order of a batch)
 *   hrf_pack_runs_sorted     humanrf/input.py:10-55 (merge_input_batches; the merged batch laid out by frame)
 *   hrf_encode4d_density_fwd humanrf/scene_representation/humanrf.py:158-186 (HumanRF.density: Decomposition4D + sigma_net + truncated_exp, one launch)
 *   hrf_encode4d_bwd_tables_binned  tcnn kernel_grid_backward x4 + compose backward, without memory-side atomics
 *   hrf_scatter_emit, hrf_scatter_accumulate  its two halves (the data-parallel step exchanges one group of temporal segments while the
 *                            next is accumulated; the reference trains on one GPU, humanrf/trainer.py:72)
 *   hrf_scatter_accumulate_signalled, hrf_stream_wait_value64  the same in ONE launch that counts each group's completion into a device
 *                            counter a stream can wait for (no counterpart in the reference: its data-parallel exchange does not exist)
 *   hrf_loss_fwd_bwd         humanrf/trainer.py:205-247, humanrf/utils/loss.py:4-10
 *   hrf_render_loss_fused    volume_rendering.py:123-145 + trainer.py:205-247 + their autograd: composite, loss and the composite's
 *                            backward of the training step in one launch
 *   hrf_adam_*               humanrf/run.py:101 (torch.optim.Adam, betas .9/.99, eps 1e-15) + GradScaler skip
 *   hrf_uniform_fill         torch.rand_like of humanrf/volume_rendering.py:63-64 (the stream hrf_prune_march draws from)
 *   hrf_occgrid_from_masks   actorshq/toolbox/native/occupancy_grid_generation.cu:16-120 (generate_from_masks)
 *   hrf_mask_dilate          actorshq/toolbox/generate_occupancy_grids_from_masks.py:64-77 (cv2.dilate of the masks)
 */
#ifndef HRF_H_
#define HRF_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hrf_stream_t; /* hipStream_t */

#define HRF_ABI_VERSION 10
#define HRF_MAX_LEVELS 16

/* Per-(segment, level) geometry of the hash grids, computed on the host. */
typedef struct hrf_level_meta {
    float scale;      /* exp2(l * log2(per_level_scale)) * base_resolution - 1 (fp32) */
    uint32_t res;     /* ceil(scale) + 1 */
    uint32_t size;    /* entries in the level */
    uint32_t offset;  /* first entry of the level, in entries, inside one encoding's table of this segment */
    uint32_t hashed;  /* 1: coherent-prime hash, 0: dense index */
} hrf_level_meta;

/* One temporal segment (humanrf.py:105-120): four tables of `entries` entries x 2 features. */
typedef struct hrf_segment_meta {
    uint64_t table_offset;   /* first entry of this segment's xyz table inside the global table buffer;
                                the xyt / yzt / xzt tables follow at +entries, +2*entries, +3*entries */
    uint32_t entries;        /* entries per encoding */
    uint32_t n_levels;
    hrf_level_meta levels[HRF_MAX_LEVELS];
} hrf_segment_meta;

const char* hrf_last_error(void);
int hrf_abi_version(void);

/* ------------------------------------------------------------------ occupancy grid ring ---- */
/* OccupanyGrid(grid_resolution, buffer_size): ring of (G,G,G) uint8 volumes [z][y][x] in HBM. */
int hrf_occgrid_create(uint64_t grid_resolution, int buffer_size, void** out_handle);
/* add_grid: copies a device volume into the next ring slot; *out_texture = opaque int64 handle consumed
 * by the sampler (it is the slot's device address). The library also builds two coarse mips (one byte per 4^3
 * block, and -- resolutions that are multiples of 16 -- one byte per widened 16^3 block; stored right behind the volume) that
 * let the march skip provably empty space, and a ray that provably sees no occupied texel leave at once, without changing any
 * result; texture handles must therefore come from hrf_occgrid_add. */
int hrf_occgrid_add(void* handle, const uint8_t* grid, uint64_t g0, uint64_t g1, uint64_t g2,
                    hrf_stream_t stream, int64_t* out_texture_host);
int hrf_occgrid_destroy(void* handle);

/* ------------------------------------------------------------------ ray sampler ------------ */
/* Stage 1 (compute_minmax_kernel, ray_sampler.cu:80-147 + light-bloom AND, :254-257):
 * for each of the R0 requested pixels: direction, [tmin,tmax], ray_mask, sample count
 * (count = mask ? (int)((tmax-tmin)/step) : 0, ray_sampler.cu:283-285).
 * light_mask may be NULL (filter_light_bloom == false). grid_textures may be NULL when !use_occupancy.
 * For a ray with mask 0 only out_mask, out_count (0) and out_dirs are defined; its out_minmax is not (the reference compacts
 * every per-ray output by the mask before anything reads it, ray_sampler.cu:258-266, and a ray that provably misses every
 * occupied texel does not walk the box to find the tmin the reference's loop would end on).
 * workspace (ABI 8; may be NULL): num_rays + 1 ints of device scratch. With it (occupancy mode, grid resolution a multiple of 16)
 * the call is two launches: a conservative test of every ray against the 16^3-block mip, then the exact march for the rays that
 * passed it, packed into full wavefronts -- nine in ten drawn rays of a training batch pass the body by. Same outputs. */
int hrf_sampler_rays(const float* inverse_krs, const float* camera_origins, const uint8_t* landscape_modes,
                     const int64_t* ray_indices, const int64_t* grid_textures, const float* aabb,
                     const uint8_t* light_mask, int64_t num_rays, int grid_resolution, int image_width,
                     int image_height, float step, int use_occupancy,
                     float* out_dirs, float* out_minmax, uint8_t* out_mask, int32_t* out_count,
                     int32_t* workspace, hrf_stream_t stream);

/* Exclusive prefix sum of n int32 (or uint8 when in_is_u8) values; out[n] receives the total
 * (out has n+1 elements). workspace: 2*ceil(n/4096)+1 ints for the multi-workgroup path (two launches; contents arbitrary), or NULL
 * (single workgroup). Device-only. Replaces the ATen cumsum / mask-compaction scans of ray_sampler.cu:254-290. */
int hrf_scan_exclusive(const void* in, int in_is_u8, int64_t n, int32_t* out, int32_t* workspace,
                       hrf_stream_t stream);

/* Boolean-mask compaction of the per-ray outputs + gathers of ray_sampler.cu:258-266.
 * slot = exclusive scan of mask. rgba_pool is the (B*P,4) uint8 pool (device resident). */
int hrf_sampler_compact_rays(const int64_t* ray_indices, const uint8_t* mask, const int32_t* slot,
                             const float* dirs_all, const float* minmax_all, const int32_t* count_all,
                             const uint8_t* rgba_pool, const float* camera_origins,
                             const int32_t* frame_numbers, const int32_t* camera_numbers,
                             int64_t num_rays_in, int64_t pixels_per_image,
                             float* out_origins, float* out_dirs, float* out_rgba, int32_t* out_frames,
                             int32_t* out_cameras, float* out_minmax, int32_t* out_count,
                             int64_t* out_ray_indices, const int32_t* cand_offset_all, int32_t* out_cand_offset,
                             hrf_stream_t stream);

/* Refill `count` pool slots from a capture that is resident in HBM, in one launch: image copy + the per-slot tables the
 * sampler reads (data_loader.py:396-511, the replacer thread's _load_and_copy_camera_frame_data). spec: DEVICE int32
 * (count, 5) = { slot, camera index in the capture, frame index in the capture, camera number, frame number };
 * capture (C, F, P, 4) uint8; all_* tables are indexed by camera number, grid_by_frame by the capture's frame index
 * (grid_textures / grid_by_frame may be NULL in aabb mode). */
int hrf_pool_replace(const int32_t* spec, int count, const uint8_t* capture, int64_t pixels_per_image,
                     int capture_frames, uint8_t* pool, const float* all_inverse_krs, const float* all_origins,
                     const uint8_t* all_landscape, const int64_t* grid_by_frame, int32_t* frame_numbers,
                     int32_t* camera_numbers, uint8_t* landscape_modes, float* inverse_krs, float* camera_origins,
                     int64_t* grid_textures, hrf_stream_t stream);

/* compute_sample_distances_kernel + final compaction (ray_sampler.cu:149-194, 322-323) over the
 * compacted rays, one wavefront per ray, ballot + prefix-popcount compaction (no host sync).
 * Pass 1 (out_t == NULL): writes out_kept[r] = surviving samples of ray r.
 * Pass 2: offsets = exclusive scan of kept; writes t and the (relative) ray index of every survivor (out_ray may be NULL).
 * Single pass (out_t AND out_kept given): offsets = exclusive scan of `count` (the candidates); ray r fills the prefix
 * [offsets[r], offsets[r] + out_kept[r]) of its candidate range -- the predicate is evaluated once.
 * num_rays_dev (may be NULL): device-side ray count when the host only knows the upper bound num_rays (lets the
 * whole sampler + prune chain run without reading the compacted ray count back); slots beyond it get kept = 0.
 * capacity: number of elements out_t / out_ray can hold (writes beyond it are dropped; the caller detects the
 * overflow from the scan total and retries with larger buffers). */
int hrf_sampler_samples(const int64_t* ray_indices, const int64_t* grid_textures, const float* origins,
                        const float* dirs, const float* minmax, const int32_t* count,
                        const int32_t* offsets, int64_t num_rays, const int32_t* num_rays_dev,
                        int64_t pixels_per_image, int grid_resolution, float step, int use_occupancy,
                        int32_t* out_kept, float* out_t, int32_t* out_ray, int64_t capacity,
                        hrf_stream_t stream);

/* ------------------------------------------------------------------ grid generation (before the path) */
/* Visual-hull carving, generate_from_masks (occupancy_grid_generation.cu:16-120): voxel (x, y, z) at
 * (x, y, z)/(G-1) - 0.5 is projected with every camera's world->pixel matrix (num_cameras x 4x4, COLUMN-major, as the
 * driver passes them, generate_occupancy_grids_from_masks.py:54-61); it is occupied (255) when at least
 * camera_coverage_threshold cameras see a non-zero mask byte in the 2x2 pixel block at the truncated projection.
 * masks: (num_cameras, width*height) uint8, row length of camera c = landscape_modes[c] ? width : height.
 * out_grid: (G, G, G) uint8 [z][y][x] -- the layout hrf_occgrid_add consumes. */
int hrf_occgrid_from_masks(const uint8_t* masks, const float* projection_matrices, const uint8_t* landscape_modes,
                           int camera_coverage_threshold, int num_cameras, int grid_resolution, int width, int height,
                           uint8_t* out_grid, hrf_stream_t stream);
/* cv2.dilate(mask, ones((k, k)), iterations=1) for num_images images of width x height (anchor (k/2, k/2), pixels
 * outside the image ignored). out must not alias masks. */
int hrf_mask_dilate(const uint8_t* masks, int width, int height, int kernel_size, int64_t num_images, uint8_t* out,
                    hrf_stream_t stream);

/* ------------------------------------------------------------------ in-repo compose op ------ */
int hrf_compose_fwd(const void* xyz_f, const void* xyt_f, const void* yzt_f, const void* xzt_f,
                    const float* vectors, const float* xyzt, int64_t n, int feature_dim, int vec_res,
                    void* out_f, hrf_stream_t stream);
int hrf_compose_bwd(const void* xyz_f, const void* xyt_f, const void* yzt_f, const void* xzt_f,
                    const float* vectors, const float* xyzt, const void* d_out, int64_t n, int feature_dim,
                    int vec_res, void* d_xyz, void* d_xyt, void* d_yzt, void* d_xzt, float* d_vectors,
                    hrf_stream_t stream);

/* ------------------------------------------------------------------ scene representation --- */
/* positions = o[ray] + t * d[ray] (volume_rendering.py:68-69,113-114), +0.5 and the frame -> (segment,
 * normalized local time) lookup (humanrf.py:159-177). jitter (may be NULL) is rand_like(t): t += jitter*step
 * is applied first and written back to t_inout (volume_rendering.py:63-64). */
int hrf_query_prep(const float* ray_origins, const float* ray_dirs, const int32_t* ray_frames,
                   const int64_t* sample_ray, float* t_inout, const float* jitter, float step,
                   const int32_t* frame_to_segment, const float* frame_to_local, int64_t n,
                   float* out_xyzt, int32_t* out_segment, hrf_stream_t stream);

/* Decomposition4D.forward for mixed segments: 4 hash-grid encodings + compose, fused.
 * tables: fp16, (entries,2) per encoding; vectors: (S,4,vec_res,32) fp32; out: (n,32) fp16. */
int hrf_encode4d_fwd(const float* xyzt, const int32_t* segment, const void* tables, const float* vectors,
                     const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                     void* out_features, void* out_enc_features, hrf_stream_t stream);

/* hrf_encode4d_fwd (with the per-encoding outputs) + hrf_density_mlp_fwd in ONE launch (ABI 8): the render pass of the fused
 * training step -- Decomposition4D.forward (decomposition4d.py:124-135) and sigma_net + truncated_exp (humanrf.py:181-186) with
 * the 64-byte feature rows handed over in LDS. Features, per-encoding features, h and sigma are bit-identical to the two calls.
 * out_h / out_sigma: one of them may be NULL. */
int hrf_encode4d_density_fwd(const float* xyzt, const int32_t* segment, const void* tables, const float* vectors,
                             const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                             void* out_features, void* out_enc_features, const void* w1, const void* w2,
                             float density_scale, void* out_h, float* out_sigma, int mlp_bf16, hrf_stream_t stream);
/* out_enc_features (may be NULL): (n,4,32) fp16, the four per-encoding outputs (xyz,xyt,yzt,xzt) that the
 * reference's autograd saves (decomposition4d.py:11); the backward needs them for the vector gradients.
 * Backward: d_features scaled by grad_scale; d_features_mode 0: (n,32) fp16, 1: (n,32) fp32, 2: fp32 level-major
 * (16,n,2) as hrf_mlp_bwd writes it for the fused training path (level-major scatter keeps the gradient tables of
 * one level cache resident); accumulates d_tables (fp32, same indexing as
 * tables, 2 floats per entry) and d_vectors (fp32) with atomics, already divided by grad_scale. Either of
 * d_tables / d_vectors may be NULL to run only the other half (data parallel: the table gradients start their
 * exchange while the vector gradients are still being computed).
 * flags (may be NULL; ABI 7): the step's found_inf flag (int32[1]), raised when a table gradient is non-finite after the half
 * gradient boundary below -- |x / grad_boundary| > 65504 is inf in the reference's half tensor, and its GradScaler then skips the
 * step (trainer.py:250-252); hrf_encode4d_bwd_tables_binned raises the same flag through its range check. */
/* grad_boundary (hrf_encode4d_bwd, hrf_encode4d_bwd_tables_binned, hrf_mlp_bwd; ABI 6): 0 = the fused backward keeps fp32 from
 * the loss to the tables. b > 0 = the reference's fp16 gradient boundaries: between its modules the gradient is a HALF tensor
 * at the GradScaler's scale (tcnn outputs are half, so autograd hands dL/d(output) over in half; the compose op's four
 * per-encoding outputs are half: decomposition4d.py:8-39, tensor_composition.cu:85-117), and only inside a tcnn module it is
 * multiplied by tcnn's loss_scale. The fused path carries `b` x the GradScaler's scale throughout (b = that loss_scale, 128):
 * with b > 0 every value that crosses such a boundary -- dL/d(sigma_net output), dL/d(features), and the compose backward's
 * d(encoding output) = v[pair] * dL/d(features) -- is replaced by half_round(x / b) * b, so a contribution below 2^-25 of
 * the GradScaler-scaled unit vanishes exactly as it does in the reference and Adam leaves such an entry alone. */
int hrf_encode4d_bwd(const float* xyzt, const int32_t* segment, const void* enc_features, const float* vectors,
                     const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                     const void* d_features, int d_features_mode, float grad_scale, float grad_boundary, float* d_tables,
                     float* d_vectors, int32_t* flags, hrf_stream_t stream);

/* The table half of hrf_encode4d_bwd (d_features_mode 2) without memory-side atomics: the same sums as tcnn's
 * kernel_grid_backward x4 + compose backward (decomposition4d.py:79-122, tensor_composition.cu:85-117), produced by a
 * radix partition -- corner gradients are aggregated along the rays in registers, appended as (entry, d_f0, d_f1)
 * records to queues private to a (tile of <= 1024 samples of one temporal segment, level, encoding, 8192-entry chunk of the
 * level table), and a second kernel accumulates every chunk in LDS in 64-bit fixed point (unit: 2^-38 of the largest
 * record of the (segment, level, encoding)) and adds it to d_tables with coalesced requests (csrc/scatter.hip says why: the chip retires ~21 G atomic requests/s and
 * the scatter needs 58 per sample; LDS integer atomics and streaming stores have no such ceiling).
 * For a batch sorted by temporal segment (hrf_pack_runs_sorted lays the training batch out by frame) no global atomic is
 * issued besides ONE coalesced add per touched entry, and d_tables is reproducible bit for bit FOR A GIVEN SAMPLE LAYOUT
 * (integer sums do not depend on the order of the records; which samples share a run of the walk, hence which partial
 * sums become records, does depend on the layout -- the collector's counting sort orders the rays of a frame by atomics,
 * so two training runs do not see the same layout). Any other order is handled (samples that sit in another segment's tile, and
 * records beyond a queue's capacity, take the direct atomic path), only slower.
 * workspace: hrf_scatter_workspace_bytes(workspace_samples, num_segments) bytes of device memory owned by these calls
 *   (one stream at a time); n <= workspace_samples; num_segments <= 1024.
 * max_level_entries: largest `size` of any level of any segment; must be <= 524288 (64 chunks: log2_hashmap_size 19 on a
 *   100-frame segment, humanrf.py:106-109) -- larger tables are served by hrf_encode4d_bwd. Tables above 65536 entries deal
 *   runs of 16 entries out to the chunks round-robin (dense levels of that size would otherwise fill the queues of a few
 *   z slabs only). flags (may be NULL): bit 0 is set when a record is non-finite or beyond the fixed-point range
 *   (the caller's found_inf flag: the optimizer then skips the step like GradScaler does). */
size_t hrf_scatter_workspace_bytes(int64_t n_samples_max, int num_segments);
int hrf_encode4d_bwd_tables_binned(const float* xyzt, const int32_t* segment, const float* vectors,
                                   const hrf_segment_meta* segments, int num_segments, int vec_res, int64_t n,
                                   const float* d_features_lm, float grad_scale, float grad_boundary, float* d_tables,
                                   void* workspace, int64_t workspace_samples, int max_level_entries, int32_t* flags,
                                   hrf_stream_t stream);
/* The two halves of hrf_encode4d_bwd_tables_binned as calls of their own (ABI 8): hrf_scatter_emit = tile table + record
 * queues, hrf_scatter_accumulate = the LDS accumulation of the temporal segments [seg_first, seg_first + seg_count) by id
 * (seg_count <= 0: every segment that owns tiles, what the combined call does). Same workspace, same stream, emit first.
 * What it is for: the reference trains on one GPU (trainer.py:72); the data-parallel step of this build accumulates the
 * segments group by group and starts each group's gradient reduce-scatter while the next group is still being accumulated
 * (SURVEY.md 8(e)). The sums do not depend on how the segments are grouped. */
int hrf_scatter_emit(const float* xyzt, const int32_t* segment, const float* vectors, const hrf_segment_meta* segments,
                     int num_segments, int vec_res, int64_t n, const float* d_features_lm, float grad_scale,
                     float grad_boundary, float* d_tables, void* workspace, int64_t workspace_samples, int max_level_entries,
                     hrf_stream_t stream);
int hrf_scatter_accumulate(const hrf_segment_meta* segments, int num_segments, float* d_tables, void* workspace,
                           int64_t workspace_samples, int max_level_entries, int32_t* flags, int seg_first, int seg_count,
                           hrf_stream_t stream);
/* ONE accumulate launch over every temporal segment by id, with completion signals per GROUP of consecutive segment ids (ABI 10; the
 * data-parallel step of this build, SURVEY.md 8(e) "overlap with the remaining backward"; no counterpart in the single-GPU reference).
 * group_bounds: n_groups (1..8) pairs (first id, last id) in HOST memory, ascending and disjoint. group_done: n_groups 64-bit counters
 * in DEVICE memory that this library only ever adds to: the accumulate grid hands its workgroups to the segments in id order, and
 * every workgroup that is done with a segment of group g adds one to group_done[g] after its sums are in d_tables -- so
 * group_done[g] grows by (segments of g) x hrf_scatter_signals_per_segment(max_level_entries) per call, and reaches that total as
 * soon as the group's table gradients are complete, while the same launch is still accumulating the groups behind it.
 * hrf_stream_wait_value64 makes a stream wait for such a total (hipStreamWaitValue64, >=): a collective issued from that stream
 * starts ~2 us after its group is complete (tools/microbench/wait_value_probe.hip), with no launch per group.
 * Enqueue the wait AFTER the launch that advances the counter (the training step does): HIP streams share a few hardware queues, and a wait
 * packet in front of that launch in the same queue would never be released.
 * hrf_can_stream_wait_value: 1 if the current device supports the wait (hipDeviceAttributeCanUseStreamWaitValue). */
int hrf_scatter_accumulate_signalled(const hrf_segment_meta* segments, int num_segments, float* d_tables, void* workspace,
                                     int64_t workspace_samples, int max_level_entries, int32_t* flags,
                                     const int32_t* group_bounds, int n_groups, uint64_t* group_done, hrf_stream_t stream);
int64_t hrf_scatter_signals_per_segment(int max_level_entries);
int hrf_can_stream_wait_value(void);
int hrf_stream_wait_value64(hrf_stream_t stream, const uint64_t* addr, uint64_t value);

/* mlp_bf16 (all MLP entry points and hrf_prune_march): 0 = weights and activations fp16 (tcnn's FullyFusedMLP, the
 * reference configuration), 1 = bf16 (BASELINE.json configs[4]): the weight pointers then hold bf16 values and every
 * rounding of an activation goes to bf16; products accumulate in fp32 on the matrix cores either way. The tensors that
 * travel between kernels (features, h, rgb, d_features) keep their fp16 / fp32 formats.
 * sigma_net + truncated_exp: features (n,32) fp16 -> h (n,16) fp16, sigma (n) fp32 = exp(h0)*density_scale.
 * w1 (64,32), w2 (16,64) 16-bit row-major (out,in) as in tcnn's params (A.2). h / sigma may be NULL. */
int hrf_density_mlp_fwd(const void* features, const void* w1, const void* w2, float density_scale,
                        int64_t n, void* out_h, float* out_sigma, int mlp_bf16, hrf_stream_t stream);

/* color_net: Composite[SH16(dir), identity(geo G + emb E)] padded with ones -> 64 -> 64 -> 16, sigmoid.
 * dirs are per ray (R,3) in [-1,1], gathered through sample_ray; h is sigma_net's output (geo = h[1:1+G]);
 * cam_emb (160,E) fp32 with per-ray camera numbers, or NULL (E = 0 or eval: zeros).
 * geometry_feature_dim G (ABI 7; model_args.py:22, 15 in the reference's configurations): 0..15 with 1 <= G + E <= 32.
 * w1 (64, in_pad), in_pad = 16 + G + E rounded up to a multiple of 16 (32 or 48: tcnn's padded input width, columns
 * [SH 16 | geo G | emb E | ones]), w2 (64,64), w3 (16,64) fp16. out_rgb (n,3) fp16.
 * n_hidden_color (ABI 9; model_args.py:31 n_hidden_layers_color, 2 in the reference's configurations): 1, 2 or 3 hidden layers of 64
 * neurons. w2 then holds the n_hidden_color - 1 hidden-to-hidden (64,64) matrices one after the other -- the order of tcnn's flat
 * parameter vector [w1 | hidden ... | w3] -- and may be NULL with one hidden layer; the gradient buffers d_cw2 / d_w2 of the backward
 * entry points have the same shape. */
int hrf_color_mlp_fwd(const float* ray_dirs, const int64_t* sample_ray, const void* h,
                      const float* cam_emb, const int32_t* ray_cameras, int emb_dim, int use_emb,
                      const void* w1, const void* w2, const void* w3, int64_t n, void* out_rgb,
                      int mlp_bf16, int geometry_feature_dim, int n_hidden_color, hrf_stream_t stream);

/* Backward of both MLPs for one batch (activations are recomputed from `features`):
 * inputs d_rgb (n,3) fp32, d_sigma (n) fp32 (both already multiplied by grad_scale by the caller's loss);
 * outputs d_features (d_features_fp32 = 0: (n,32) fp16, 1: (n,32) fp32, 2: fp32 level-major (16,n,2); scaled), and fp32 weight gradients ACCUMULATED (atomics) into
 * d_sw1,d_sw2,d_cw1,d_cw2,d_cw3 (same shapes as the weights), d_cam_emb (160,E) -- all still scaled.
 * flags[0] is set to 1 if any fp16 conversion overflowed (GradScaler found_inf). */
int hrf_mlp_bwd(const void* features, const float* ray_dirs, const int64_t* sample_ray,
                const float* cam_emb, const int32_t* ray_cameras, int emb_dim, int use_emb,
                const void* sw1, const void* sw2, const void* cw1, const void* cw2, const void* cw3,
                float density_scale, const float* d_rgb, const float* d_sigma, int64_t n,
                void* d_features, int d_features_fp32, float grad_boundary, float* d_sw1, float* d_sw2, float* d_cw1,
                float* d_cw2, float* d_cw3, float* d_cam_emb, int32_t* flags, int mlp_bf16, int geometry_feature_dim,
                int n_hidden_color, hrf_stream_t stream);
/* The two networks differentiated separately -- the backward passes of tcnn.Network (sigma_net) and
 * tcnn.NetworkWithInputEncoding (color_net) as stand-alone modules (humanrf.py:123-156; humanrf_amd.compat.tinycudann).
 * hrf_density_mlp_bwd: d_h (n,16) fp32 = gradient of sigma_net's 16 outputs (scaled by the caller like d_rgb / d_sigma
 *   above) -> d_features (n,32) fp16 (d_features_fp32 = 0), fp32 (1) or fp32 level-major (16,n,2) (2: what the table scatter
 *   of the fused training path reads); grad_boundary as in hrf_mlp_bwd; d_w1 (64,32), d_w2 (16,64) accumulated (+=).
 * hrf_color_mlp_bwd: inputs as hrf_color_mlp_fwd, d_rgb (n,3) fp32 -> d_h (n,16) fp32 = gradient of the geometry input;
 *   row 0, the density logit the colour network does not read, is zero -- or, when d_sigma (n) fp32 is given (may be NULL),
 *   d_sigma * density_scale * exp(clamp(h[0], -15, 15)), the backward of truncated_exp (activation.py:23-39): d_h is then
 *   the whole upstream gradient of sigma_net, and the pair hrf_color_mlp_bwd + hrf_density_mlp_bwd equals hrf_mlp_bwd with
 *   each kernel at two wavefronts per SIMD (the fused kernel's 176 accumulator registers hold it to one).
 *   d_w1 / d_w2 / d_w3 / d_cam_emb accumulated.
 * flags bit 0: a 16-bit intermediate overflowed (the caller's found_inf). */
int hrf_density_mlp_bwd(const void* features, const void* w1, const void* w2, const float* d_h, int64_t n,
                        void* d_features, int d_features_fp32, float grad_boundary, float* d_w1, float* d_w2,
                        int32_t* flags, int mlp_bf16, hrf_stream_t stream);
int hrf_color_mlp_bwd(const float* ray_dirs, const int64_t* sample_ray, const void* h, const float* cam_emb,
                      const int32_t* ray_cameras, int emb_dim, int use_emb, const void* w1, const void* w2,
                      const void* w3, const float* d_rgb, const float* d_sigma, float density_scale, int64_t n,
                      float* d_h, float* d_w1, float* d_w2, float* d_w3, float* d_cam_emb, int32_t* flags, int mlp_bf16,
                      int geometry_feature_dim, int n_hidden_color, hrf_stream_t stream);


/* ------------------------------------------------------------------ volume rendering ------- */
/* ray_start[r] = first sample of ray r in the sorted sample_ray array (ray_start[R] = n). */
int hrf_ray_offsets(const int64_t* sample_ray, int64_t n, int64_t num_rays, int32_t* out_ray_start,
                    hrf_stream_t stream);

/* render_visibility (early_stop_eps, alpha_thre): sequential fp32 transmittance product per ray.
 * alphas (n) as the reference passes them (volume_rendering.py:76), or NULL to compute
 * alpha = 1 - exp(-sigma*step) in-kernel; out_vis (n) uint8, out_kept[r] = visible samples of ray r
 * (may be NULL). */
int hrf_visibility(const float* alphas, const float* sigma, const int32_t* ray_start, int64_t num_rays,
                   float step, float early_stop_eps, float alpha_thre, uint8_t* out_vis, int32_t* out_kept,
                   hrf_stream_t stream);

/* Fused pruning pass (prune_samples, volume_rendering.py:63-84 + HumanRF.density, humanrf.py:158-186): one
 * wavefront marches one ray through its run [ray_start[r], ray_start[r+1]) of t0, 64 samples per step --
 * jitter, position, 4D hash encoding, sigma_net, alpha, sequential transmittance, visibility -- and stops after
 * the chunk in which T drops below early_stop_eps (later samples are invisible by the prefix property, so the
 * result equals hrf_visibility over all samples). Survivors of ray r are written to t_stage[ray_start[r] + k],
 * k < ray_cnt[r] (sigma_stage likewise, may be NULL); ray_evaluated (may be NULL) counts encoded samples.
 * hrf_pack_runs then packs the ranges: out_offset = exclusive scan of ray_cnt; ray_base is added to the ray
 * indices it writes (merging of batches, humanrf/input.py:24-31). num_rays_dev as in hrf_sampler_samples.
 * ray_order (may be NULL) is a schedule, not a result: the ray ids sorted by a per-frame key in [0, num_keys),
 * num_keys <= 1024 -- the temporal segment, or the rank of the frame (finer: rays of one frame also share the time
 * slice of the xyt / yzt / xzt tables) -- built by hrf_ray_segment_order (workspace = 2*num_keys int32). The march
 * hands the k-th eighth of that order to the k-th XCD so that each 4 MB L2 holds the tables of one or two
 * segments / frames instead of all of them. Outputs do not depend on it.
 * ray_len (may be NULL): the run of ray r is [ray_start[r], ray_start[r] + ray_len[r]) instead of ending at
 * ray_start[r+1] (single-pass hrf_sampler_samples stages by candidate count). jitter_seed (0 = none; exclusive with
 * `jitter`): the jitter of staged sample i is value i of the counter-based stream hrf_uniform_fill(jitter_seed) writes,
 * computed in the kernel. totals (may be NULL): uint64[2], += samples handed to the pass, += samples encoded. */
int hrf_ray_segment_order(const int32_t* ray_frames, const int32_t* frame_to_key, int64_t num_rays,
                          const int32_t* num_rays_dev, int num_keys, int32_t* workspace, int32_t* out_order,
                          hrf_stream_t stream);
/* The same order, carrying one int32 per ray along: out_values[i] = values[out_order[i]] (values / out_values NULL: plain
 * hrf_ray_segment_order). Used with the per-ray visible-sample counts to lay the training batch out in frame order. */
int hrf_ray_segment_order_values(const int32_t* ray_frames, const int32_t* frame_to_key, int64_t num_rays,
                                 const int32_t* num_rays_dev, int num_keys, int32_t* workspace, int32_t* out_order,
                                 const int32_t* values, int32_t* out_values, hrf_stream_t stream);
int hrf_prune_march(const float* ray_origins, const float* ray_dirs, const int32_t* ray_frames,
                    const int32_t* ray_start, const float* t0, const float* jitter, float step,
                    float early_stop_eps, float alpha_thre, const int32_t* frame_to_segment,
                    const float* frame_to_local, const void* tables, const float* vectors,
                    const hrf_segment_meta* segments, int num_segments, int vec_res, const void* w1,
                    const void* w2, float density_scale, int64_t num_rays, const int32_t* num_rays_dev,
                    int64_t capacity, float* t_stage, float* sigma_stage, int32_t* ray_cnt, int32_t* ray_evaluated,
                    const int32_t* ray_order, const int32_t* ray_len, uint32_t jitter_seed, uint64_t* totals,
                    int mlp_bf16, hrf_stream_t stream);
/* The batch-growing loop of Trainer.train (trainer.py:138-163) replayed on the device over speculatively marched rays:
 * slot = exclusive scan of the ray mask over the drawn rays of a prefetched set, out_offset = exclusive scan of ray_cnt
 * over the compacted rays marched from ray_base on. Loop state in: drawn rays already used, next batch size r0, totals so
 * far; the loop runs while used + r0 <= spec_end (the drawn rays marched). plan: int64[16] = { done, iterations run,
 * drawn rays used, next r0, compacted rays up to `used`, visible samples of this chunk, error, total drawn rays,
 * *extra (any device int32 the caller wants in the same read-back; extra may be NULL), slot[spec_end],
 * out_offset at 1/4, 2/4, 3/4 of the chunk's compacted rays (ray-aligned cut points of the batch), those rays, 0, 0 }. */
int hrf_batch_plan(const int32_t* slot, const int32_t* out_offset, int64_t ray_base, int64_t used, int64_t spec_end,
                   int64_t r0, int64_t total_rays, int64_t total_samples, int64_t samples_max, const int32_t* extra,
                   int64_t* plan, hrf_stream_t stream);
int hrf_pack_runs(const int32_t* ray_start, const int32_t* ray_cnt, const int32_t* out_offset,
                  const float* t_stage, int64_t num_rays, const int32_t* num_rays_dev, int64_t ray_base,
                  float* out_t, int64_t* out_ray, hrf_stream_t stream);
/* hrf_pack_runs in a given ray order (merge_input_batches, humanrf/input.py:10-55, for a batch laid out by frame): sorted
 * ray i is ray order[i]; its survivors go to out_t / out_ray at out_offset_sorted[i] (exclusive scan of ray_cnt in sorted
 * order) with ray id i, and its per-ray record (origin, direction, rgba, frame, camera, minmax, optional pixel id) is copied
 * to row i of the o_* arrays. The batch is a set of i.i.d. rays: rendering, the loss means and the gradient sums do not
 * depend on the order; in frame order consecutive samples read one temporal segment's tables. */
int hrf_pack_runs_sorted(const int32_t* order, const int32_t* ray_start, const int32_t* ray_cnt,
                         const int32_t* out_offset_sorted, const float* t_stage, int64_t num_rays,
                         const float* origins, const float* dirs, const float* rgba, const int32_t* frames,
                         const int32_t* cams, const float* minmax, const int64_t* pixel, float* o_origins,
                         float* o_dirs, float* o_rgba, int32_t* o_frames, int32_t* o_cams, float* o_minmax,
                         int64_t* o_pixel, float* out_t, int64_t* out_ray, hrf_stream_t stream);

/* Stand-alone forms of nerfacc 0.3.1's render_weight_from_density and accumulate_along_rays (volume_rendering.py:
 * 123-141) for callers written against those functions; the training step uses hrf_composite_* (fused). Samples sorted
 * by ray, ray_start = hrf_ray_offsets. values (n, value_dim) fp32 or NULL (weights only, value_dim 1). */
int hrf_weights_fwd(const float* sigma, const float* t_starts, const float* t_ends, const int32_t* ray_start,
                    int64_t num_rays, float* out_weights, hrf_stream_t stream);
int hrf_weights_bwd(const float* sigma, const float* t_starts, const float* t_ends, const int32_t* ray_start,
                    const float* d_weights, int64_t num_rays, float* d_sigma, hrf_stream_t stream);
int hrf_accumulate_fwd(const float* weights, const float* values, int value_dim, const int32_t* ray_start,
                       int64_t num_rays, float* out, hrf_stream_t stream);
int hrf_accumulate_bwd(const float* weights, const float* values, int value_dim, const int64_t* sample_ray,
                       const float* d_out, int64_t n, float* d_weights, float* d_values, hrf_stream_t stream);

/* Boolean-mask compaction of per-sample arrays (volume_rendering.py:83-84): slot = exclusive scan of vis. */
int hrf_compact_samples(const uint8_t* vis, const int32_t* slot, const float* t, const int64_t* sample_ray,
                        int64_t n, float* out_t, int64_t* out_sample_ray, hrf_stream_t stream);

/* render_weight_from_density + accumulate_along_rays x2 + background blend. rgb (n,3) fp16.
 * background (R,3) fp32 or NULL. Saves nothing: backward recomputes. */
int hrf_composite_fwd(const float* sigma, const void* rgb, const float* t, const int32_t* ray_start,
                      const float* background, int64_t num_rays, float step, float* out_color,
                      float* out_acc, hrf_stream_t stream);
int hrf_composite_bwd(const float* sigma, const void* rgb, const float* t, const int32_t* ray_start,
                      const float* background, const float* d_color, const float* d_acc, int64_t num_rays,
                      float step, float* d_sigma, float* d_rgb, hrf_stream_t stream);

/* torch.amp.GradScaler's state, on the DEVICE (trainer.py:74, 250-252): the loss kernel multiplies the loss gradient by
 * `scale`, hrf_adam_multi divides by it and then applies GradScaler.update(): backoff on a non-finite gradient, growth
 * after growth_interval consecutive clean steps. No host synchronisation is involved, as in torch. */
typedef struct hrf_grad_scaler {
    float scale;              /* torch default init_scale 65536 */
    float growth_factor;      /* 2 */
    float backoff_factor;     /* 0.5 */
    int32_t growth_interval;  /* config.training.scaler_growth_interval (example_humanrf.py:22: 100000) */
    int32_t growth_tracker;   /* consecutive clean steps since the last change */
    int32_t reserved[3];
} hrf_grad_scaler;

/* Huber(delta)+bce_weight*BCE loss and its gradient w.r.t. color / acc, times grad_scale (times scaler->scale when a
 * scaler is given). norm_rays: the number of rays the means are taken over -- 0 = num_rays; a caller that feeds one
 * batch in several pieces passes the batch's ray count with every piece.
 * out_sums[0] += sum huber, [1] += sum bce, [2] += sum squared error (for PSNR).
 * group_touched (may be NULL, then ray_frames / frame_to_segment are unused): group_touched[1 + segment of ray r] = 1
 * for every ray of the batch -- the segments whose parameters receive a gradient in the reference (humanrf.py:159-163),
 * consumed by hrf_adam_multi. */
int hrf_loss_fwd_bwd(const float* color, const float* acc, const float* rgba, const float* background,
                     int64_t num_rays, int64_t norm_rays, float huber_delta, float bce_weight, float grad_scale,
                     float* d_color, float* d_acc, float* out_sums, const int32_t* ray_frames,
                     const int32_t* frame_to_segment, int32_t* group_touched, const hrf_grad_scaler* scaler,
                     hrf_stream_t stream);

/* hrf_composite_fwd + hrf_loss_fwd_bwd + hrf_composite_bwd in ONE launch (ABI 8; one wavefront per ray; the same expressions in
 * the same order: colour, opacity, d_sigma and d_rgb are bit-identical to the three calls). What it replaces in the reference:
 * nerfacc's render_weight_from_density / accumulate_along_rays and the background blend (volume_rendering.py:123-145), the
 * losses (trainer.py:205-247, utils/loss.py:4-10) and autograd's backward through them. out_color / out_acc may be NULL.
 * out_sums (may be NULL): as hrf_loss_fwd_bwd (a wavefront walks several rays and a workgroup adds its total once). */
int hrf_render_loss_fused(const float* sigma, const void* rgb, const float* t, const int32_t* ray_start,
                          const float* background, const float* rgba, int64_t num_rays, int64_t norm_rays, float step,
                          float huber_delta, float bce_weight, float grad_scale, const hrf_grad_scaler* scaler,
                          const int32_t* ray_frames, const int32_t* frame_to_segment, int32_t* group_touched,
                          float* out_color, float* out_acc, float* d_sigma, float* d_rgb, float* out_sums,
                          hrf_stream_t stream);

/* One stand-alone tcnn HashGrid encoding, as decomposition4d.py:79-122 instantiates it (tcnn.Encoding, 3 input dims): for
 * code written against tinycudann's modules (humanrf_amd.compat.tinycudann); the training path uses hrf_encode4d_*.
 * x (n,3) fp32 in [0,1]; table: the encoding's entries as __half2; meta: ONE hrf_segment_meta describing its levels
 * (table_offset / entries unused); out (n, n_levels*2) __half. Backward: d_table fp32 (entries*2), += weight * d_features /
 * grad_scale; d_features (n, n_levels*2) __half, or fp32 when d_features_fp32 != 0. */
int hrf_hashgrid_fwd(const float* x, const void* table, const hrf_segment_meta* meta, int n_levels, int64_t n,
                     void* out_features, hrf_stream_t stream);
int hrf_hashgrid_bwd(const float* x, const hrf_segment_meta* meta, int n_levels, int64_t n, const void* d_features,
                     int d_features_fp32, float grad_scale, float* d_table, hrf_stream_t stream);

/* ------------------------------------------------------------------ optimizer -------------- */
/* torch.optim.Adam step (no weight decay / amsgrad) on fp32 master params; grads are divided by
 * grad_scale first; optionally refreshes the fp16 copy used by the kernels (p16 may be NULL) and zeroes
 * the gradient. Skipped entirely when flags[0] != 0 (GradScaler found_inf semantics), in which case only
 * the gradient is zeroed. bias corrections are passed precomputed: bc1 = 1-beta1^t, bc2 = 1-beta2^t. */
int hrf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* p16, int64_t n,
                  float lr, float beta1, float beta2, float eps, float bc1, float bc2, float grad_scale,
                  const int32_t* flags, hrf_stream_t stream);

/* The same step for every parameter tensor of the model in ONE launch, optimizer bookkeeping on the device, with
 * torch.optim.Adam's per-parameter behaviour under the reference's training loop: parameters without a gradient are
 * skipped and keep their own step count (the reference only runs the temporal segments a batch touches,
 * humanrf.py:159-179, and zero_grad(set_to_none=True), trainer.py:174), found_inf skips everything (trainer.py:250-252).
 * tensors: DEVICE array of `count` descriptors; `group` 0 = always stepped, g > 0 = stepped when touched[g] != 0.
 * state: DEVICE int32[4 + 2*num_groups] = { found_inf of this step, steps skipped, internal, unused,
 *   steps[num_groups] (Adam's t per group), touched[num_groups] }. The kernel advances steps, clears found_inf and the
 *   touched flags. max_elements: upper bound of the parameters one launch may step (sizes the grid).
 *   Gradients are divided by grad_scale (times scaler->scale when a scaler is given; the scaler is then updated).
 *   workspace: DEVICE scratch of hrf_adam_workspace_bytes() bytes (the list of tensors this launch steps). */
typedef struct hrf_adam_tensor {
    float* param;    /* NULL: a gradient range this rank does not own (reduce-scatter + sharded Adam): it is only zeroed */
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    void* p16;       /* 16-bit copy to refresh, or NULL */
    int64_t n;
    int32_t group;
    int32_t reserved; /* bit 0: the 16-bit copy is bf16 (MLP weights of a bf16 model), else fp16 */
} hrf_adam_tensor;
size_t hrf_adam_workspace_bytes(void);
int hrf_adam_multi(const hrf_adam_tensor* tensors, int count, int num_groups, int64_t max_elements, float lr,
                   float beta1, float beta2, float eps, float grad_scale, int32_t* state, hrf_grad_scaler* scaler,
                   void* workspace, hrf_stream_t stream);

/* out[i] = value i of the counter-based uniform [0,1) stream `seed` (24 random bits, like torch.rand): the numbers
 * hrf_prune_march draws in-kernel for jitter_seed == seed. n < 2^32. */
int hrf_uniform_fill(uint32_t seed, int64_t n, float* out, hrf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HRF_H_ */
