/*
 * pch_hip.h -- C ABI of libpch_hip.so: the MI355X (gfx950) implementation of the
 * pointcloudhookup ground-removal + tower-clustering hot path.
 *
 * The reference (Daniel-Starr/pointcloudhookup) is pure Python; the arithmetic of its
 * hot path lives in library calls (Open3D, numpy, scikit-learn).  Every entry point
 * below replaces exactly one of those call sites (cited as reference file:line), so the
 * reference's Python modules can bind them with ctypes.
 *
 * Conventions
 *  - plain C: pointers, sizes, scalars.  No torch/HIP types in signatures
 *    (`stream` is a hipStream_t passed as void*; NULL = default stream).
 *  - every `const T*` / `T*` argument is a DEVICE pointer unless its name ends in
 *    `_host`.  Buffers are caller-owned.  Every entry point looks up the device that owns
 *    its first buffer (hipPointerGetAttributes), makes it current for the call
 *    (hipSetDevice) and restores the caller's device on return; a host pointer where a
 *    device pointer is expected gives PCH_ERR_ARG.
 *  - return value: 0 = ok, <0 = PCH_ERR_*.  pch_last_error() gives a thread-local text.
 *  - the library never allocates device memory: scratch is a caller-provided workspace
 *    whose size comes from the matching *_ws_bytes() (a pure host function).
 *  - entry points enqueue work on `stream` and return without synchronising, except the
 *    ones documented "synchronises" (they must read a device-side count to size the next
 *    launch).
 *  - results are bit-exact restatements of the reference's arithmetic: the library is
 *    built with -ffp-contract=off; float64 division is IEEE.
 */
#ifndef PCH_HIP_H
#define PCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCH_VERSION 300

#define PCH_OK               0
#define PCH_ERR_ARG         -1   /* bad argument (null pointer, negative size, eps<=0 ...) */
#define PCH_ERR_WORKSPACE   -2   /* workspace too small */
#define PCH_ERR_HIP         -3   /* a HIP runtime call failed */
#define PCH_ERR_RANGE       -4   /* grid does not fit the 64-bit cell/voxel key */
#define PCH_ERR_NODEVICE    -5   /* no gfx950 device visible */
#define PCH_ERR_TIMEOUT     -6   /* a device-side wait between workgroups ran out of its budget (see below) */

/* Device-side waits are bounded.  The order-preserving compactions (pch_ground_filter_f32, pch_filter_gt_f32,
 * pch_crop_aabb_f64) and the voxel finisher (pch_voxel_downsample_f64) chain their workgroups by a single-pass
 * look-back: a workgroup polls the status words of the workgroups in front of it.  Every such poll loop has a
 * wall-clock budget (4 s); a workgroup that exceeds it gives up, the kernel drains, and the call's count word
 * (*out_count / *out_m, device memory) reads NEGATIVE instead of holding a count - the outputs are then
 * undefined.  Entry points that read the count on the host themselves (pch_tower_clusters_f32) return
 * PCH_ERR_TIMEOUT; callers of the asynchronous entry points must test the sign when they read the count
 * (pointcloudhookup_amd/ops.py does and raises).  Nothing in normal operation comes near the budget: it exists
 * so that a GPU shared with other processes can never be left with a resident spinning grid. */

int         pch_version(void);
const char* pch_last_error(void);
/* number of visible HIP devices (0 if none); does not initialise a context */
int         pch_device_count(void);

/* ------------------------------------------------------------------ stage A
 * Open3D PointCloud.voxel_down_sample applied per file-order chunk.
 * Replaces: ui/import_PC.py:8-13 (process_chunk) inside the loop ui/import_PC.py:45-58
 *           (twin: ui/Sampling.py:10-18,46-60).
 * xyz        [n,3] float64, row-major
 * chunk_size points per chunk (<=0: one chunk); every chunk has its own grid origin
 *            min(chunk) - voxel/2; duplicates across chunks are kept (as the reference).
 * out_idx    [n,3] int32   voxel index triplets   } capacity n rows, the first *out_m rows
 * out_mean   [n,3] float64 sum(points in order)/count } are valid; rows are grouped by chunk
 * out_count  [n]   int32                           } (order inside a chunk: see below)
 * Order inside a chunk: Open3D emits unordered_map iteration order, i.e. unspecified; parity is the SET of
 * (index, mean, count) per chunk.  This library emits coarse grid cells (the top bits of [ix|iy|iz]) in ascending
 * order and, inside a cell, the voxels in the order of their first points - or sorted by (ix,iy,iz) where a cell
 * takes the sorting path (dense cells, wide keys).  Deterministic for a given input.
 * out_chunk_offsets [nchunks+1] int64 (may be NULL): slice of each chunk in the output
 * out_m      [1] int64 number of voxels
 */
size_t pch_voxel_downsample_ws_bytes(int64_t n, int64_t chunk_size);
int pch_voxel_downsample_f64(const double* xyz, int64_t n, double voxel_size,
                             int64_t chunk_size,
                             int32_t* out_idx, double* out_mean, int32_t* out_count,
                             int64_t* out_chunk_offsets, int64_t* out_m,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ LAS files (native reader / writer)
 * What laspy does for the reference on the hot path, done by the library itself (host C++ + one decode
 * kernel): read the public header block; read the X, Y, Z integers of a record range straight into a
 * DEVICE buffer (memory-mapped file -> pinned double buffer -> H2D copy overlapped with the decode);
 * write a LAS file whose records carry X, Y, Z and zeros elsewhere (what laspy writes for a LasData whose
 * only assigned dimensions are x, y, z).  LAS 1.0-1.4, point formats 0-10, uncompressed; VLRs, extra bytes
 * and padding in front of the point data are skipped through header_size / offset_to_points /
 * record_length.
 * Replaces: laspy.read(...) / laspy.open(...).read() (ui/import_PC.py:28, utils/tower_extraction.py:60-61,
 *           ui/extract.py:114-115); LasHeader / LasData(...).write (ui/import_PC.py:35-42,64-65,
 *           utils/tower_extraction.py:243-257).  Both synchronise `stream` before returning. */
typedef struct PchLasHeader {
    uint8_t  version_major, version_minor, point_format, reserved0;
    uint16_t header_size, record_length;
    uint32_t offset_to_points, num_vlrs;
    uint64_t point_count;
    double   scales[3], offsets[3], mins[3], maxs[3];
} PchLasHeader;
int    pch_las_read_header(const char* path_host, PchLasHeader* out_host);
size_t pch_las_read_ws_bytes(void);
/* out_XYZ [count,3] int32 (device): records first .. first+count-1;  ws: device scratch */
int    pch_las_read_xyz_i32(const char* path_host, int64_t first, int64_t count, int32_t* out_XYZ,
                            void* ws, size_t ws_bytes, void* stream);
/* hdr_host: point_format, version, scales, offsets are used; XYZ [n,3] int32 (device) */
int    pch_las_write_xyz_i32(const char* path_host, const PchLasHeader* hdr_host, const int32_t* XYZ,
                             int64_t n, void* stream);

/* LAS point records -> X,Y,Z: gathers the three little-endian int32 at the start of every
 * `record_len`-byte record (LAS 1.0-1.4 point formats 0-10 all start with X,Y,Z).
 * Replaces: laspy's record parsing behind las.X/.Y/.Z (ui/import_PC.py:28,
 *           utils/tower_extraction.py:60-61).  records: raw bytes [n*record_len] on the device. */
int pch_las_records_xyz_i32(const uint8_t* records, int64_t n, int32_t record_len,
                            int32_t* out_XYZ, void* stream);

/* laspy scaled view, stage A0: out = (double)X * scale + offset (no FMA).
 * Replaces: the chunk.x/.y/.z reads at ui/import_PC.py:47-48 and las.x/.y/.z at
 *           utils/tower_extraction.py:62.  XYZ [n,3] int32 (LAS record ints, AoS). */
int pch_las_scale_i32_f64(const int32_t* XYZ, int64_t n, const double* scale3_host,
                          const double* offset3_host, double* out_xyz, void* stream);
/* laspy coordinate setter, stage A3: out = (int32) rint((v - offset) / scale).
 * Replaces: downsampled.x/.y/.z = ... at ui/import_PC.py:61-63 and
 *           utils/tower_extraction.py:254-256. */
int pch_las_unscale_f64_i32(const double* xyz, int64_t n, const double* scale3_host,
                            const double* offset3_host, int32_t* out_XYZ, void* stream);

/* ------------------------------------------------------------------ stage B
 * .astype(np.float32) of utils/tower_extraction.py:62 */
int pch_cast_f64_f32(const double* in, int64_t count, float* out, void* stream);

/* np.mean(raw_points, axis=0) for a C-order [n,3] float32 array, bit-exact: a
 * SEQUENTIAL float32 running sum per column divided by float32(n).
 * Replaces: utils/tower_extraction.py:63.   out_centroid [3] float32. */
size_t pch_mean_seq_f32_ws_bytes(int64_t n);
int pch_mean_seq_f32(const float* xyz, int64_t n, float* out_centroid,
                     void* ws, size_t ws_bytes, void* stream);
/* One file-order SHARD of that sequential sum (config 4: every rank holds a consecutive slice of the rows and the
 * three running sums travel down the line, 12 bytes per hop): continues the float32 running sums sum_in3 (device
 * float[3]; NULL = +0.0, the first shard) over these n rows, exactly as numpy's loop would.
 *   total_n == 0: out3 = the running sums AFTER the rows (hand them to the next shard);
 *   total_n  > 0: out3 = running sums / float32(total_n) - the centroid of all total_n rows (last shard).
 * Shards chained in file order give np.mean(concatenation, axis=0) bit for bit; n == 0 passes the sums through.
 * phase: 0 = everything; 1 = only the summary tables of these rows (they do not depend on sum_in3: every rank
 * builds them at once, sum_in3 / out3 are ignored); 2 = only the short serial walk over the tables a phase-1 call
 * left in the SAME workspace (the caller keeps that workspace untouched in between) - so that along a chain of
 * shards only the walks (~0.1-0.3 ms each) are serial, not the passes over the rows.
 * zcol_out (optional, [n] float32): receives a contiguous copy of the z column, written by the table pass that
 * reads the rows anyway (phases 0 and 1) - what the percentile passes of stage B2 then read instead of the rows.
 * Workspace: pch_mean_seq_f32_ws_bytes(n).  Replaces: utils/tower_extraction.py:63 on a sharded array. */
int pch_mean_seq_partial_f32(const float* xyz, int64_t n, const float* sum_in3, int64_t total_n,
                             float* out3, float* zcol_out, int32_t phase, void* ws, size_t ws_bytes, void* stream);
/* same result from one workgroup adding element by element (O(n) serial; kept only to
 * cross-check the parallel algorithm above) */
int pch_mean_seq_serial_f32(const float* xyz, int64_t n, float* out_centroid, void* stream);

/* np.percentile(v, q) (method 'linear', numpy 2.x float32 semantics) of the strided
 * float32 column v[i] = base[i*stride] - (sub ? *sub : 0).
 * Replaces: utils/tower_extraction.py:82-83 (z_values = points[:,2]; percentile 25).
 * out [1] float32. */
size_t pch_percentile_f32_ws_bytes(int64_t n);
int pch_percentile_f32(const float* base, int64_t n, int64_t stride, const float* sub,
                       double q_percent, float* out,
                       void* ws, size_t ws_bytes, void* stream);

/* The three histogram passes of the radix select and the "smallest key above" pass one by one, for a
 * percentile over values that are spread over several GPUs (pointcloudhookup_amd/tiles.py::shared_percentile:
 * every rank histograms its part, the histograms are all-reduced, the host picks the bin).  Keys are the
 * order-preserving uint32 images of the floats (sign bit flipped / complemented; NaN = 0xFFFFFFFF).
 * pass 0: bins = key >> 20 (4096), pass 1: (key >> 8) & 0xFFF among keys with key >> 20 == prefix,
 * pass 2: key & 0xFF among keys with key >> 8 == prefix.  out_hist [4096] uint32 (device), out_nan [1] uint64
 * (device, may be NULL; filled by pass 0).  ws: pch_percentile_f32_ws_bytes.  Both synchronise. */
int pch_select_hist_f32(const float* base, int64_t n, int64_t stride, int32_t pass, uint32_t prefix,
                        uint32_t* out_hist, unsigned long long* out_nan, void* ws, size_t ws_bytes, void* stream);
int pch_select_min_above_f32(const float* base, int64_t n, int64_t stride, uint32_t key, uint32_t* out_key,
                             void* ws, size_t ws_bytes, void* stream);

/* keep = (z - centroid[2]) > threshold, points = raw - centroid, order preserving: the sweep of the fused filter
 * with GIVEN centroid and threshold (utils/tower_extraction.py:64,84 once both are known - the shared values of
 * a tiled run).  out_points [n,3] capacity, out_index [n] int32 (may be NULL), out_count [1] int64, out_aabb [6]
 * float32 (may be NULL).  Synchronises. */
size_t pch_filter_gt_ws_bytes(int64_t n);
int pch_filter_gt_f32(const float* raw, int64_t n, const float* centroid3_host, float threshold,
                      float* out_points, int32_t* out_index, int64_t* out_count, float* out_aabb,
                      void* ws, size_t ws_bytes, void* stream);

/* Fused stage B: centroid, centring, percentile threshold, order-preserving compaction.
 * Replaces: utils/tower_extraction.py:63-64,82-89
 *   centroid = mean(raw); points = raw - centroid; base = percentile(points[:,2], pct);
 *   keep = z > base + offset; if kept < min_keep: keep = z > base + fallback_offset.
 * raw          [n,3] float32
 * out_points   [n,3] float32 capacity; first *out_count rows = points[keep] (file order)
 * out_index    [n]   int32 (may be NULL): original row of each kept point
 * out_scalars  [8]   float32: centroid xyz, base, threshold, used_fallback(0/1),
 *                    [6] = the BITS of the uint32 count kept at `offset` (an integer in a
 *                    float slot - reinterpret, do not convert), [7] = 0
 * out_count    [1]   int64
 * out_aabb     [6]   float32 (may be NULL): min xyz, max xyz of the kept points
 */
size_t pch_ground_filter_ws_bytes(int64_t n);
int pch_ground_filter_f32(const float* raw, int64_t n, double pct, float offset,
                          float fallback_offset, int64_t min_keep,
                          float* out_points, int32_t* out_index, float* out_scalars,
                          int64_t* out_count, float* out_aabb,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ stage C
 * sklearn.cluster.DBSCAN(eps, min_samples, algorithm='ball_tree').fit(chunk).labels_ for
 * every consecutive chunk_size-row chunk, with the reference's label offsetting.
 * Replaces: the loop utils/tower_extraction.py:96-117.
 * xyz        [n,3] float32 (n known on the host)
 * chunk_size rows per fit (reference: 50000); <=0: one global fit
 * labels     [n] int32: -1 noise, else cluster id, ids dense and ordered exactly as the
 *            reference's all_labels
 * core       [n] uint8 (may be NULL): 1 for core samples
 * out_nclusters [1] int32
 * A chunk that contains NaN/inf is left at -1 and does not advance the label counter (sklearn
 * raises ValueError for it and the reference catches that, utils/tower_extraction.py:118-119).
 * Exact rule implemented (equals sklearn's sweep): neighbours are
 * sum_j((double)x_j-(double)y_j)^2 <= eps*eps; core = >= min_samples neighbours incl.
 * self; clusters = components of the core graph numbered by smallest core index;
 * border point -> smallest cluster id among its core neighbours.
 * Synchronises once (reads the bounding box to size the cell grid) unless aabb_host
 * (min xyz, max xyz of the input, any superset box) is given.
 * Cell sort: one workgroup per chunk or one global radix sort, chosen by the chunk count;
 * environment PCH_DBSCAN_SORT=chunk|global (read once per process) or
 * pch_dbscan_set_sort_mode(0 auto | 1 chunk | 2 global) force one - same results either way.
 */
void   pch_dbscan_set_sort_mode(int mode);
size_t pch_dbscan_ws_bytes(int64_t n);
int pch_dbscan_f32(const float* xyz, int64_t n, double eps, int32_t min_samples,
                   int64_t chunk_size, const float* aabb_host,
                   int32_t* labels, uint8_t* core, int32_t* out_nclusters,
                   void* ws, size_t ws_bytes, void* stream);

/* Relabels the result of the pch_dbscan_f32 call this thread made last, on the SAME workspace (which still
 * holds the cell grid, the core flags and the sorted points): core points take map[old cluster id],
 * every non-core point is assigned again as "smallest NEW id among its core neighbours within eps, else
 * -1" (border points must be re-decided: the order of the ids may have changed).  Used by the cross-tile
 * reconciliation (pointcloudhookup_amd/tiles.py): clusters cut by a tile edge are united and renumbered
 * globally, which generalises the per-chunk label offsets of utils/tower_extraction.py:113-116.
 * map [nmap] int32 (device): new id per old id (-1 = drop).  labels [n] int32: in/out.
 * Workspace rule: the library remembers, per thread, the workspace of the last pch_dbscan_f32 call; ANY pch_*
 * call of that thread whose workspace overlaps it forgets it again, and this call then returns PCH_ERR_ARG
 * ("must follow pch_dbscan_f32 ... untouched workspace") rather than following overwritten indices.  Give the
 * fit a workspace of its own if other calls run between the fit and the relabel (ops.DbscanFit does). */
int pch_dbscan_relabel_i32(const int32_t* map, int32_t nmap, int64_t n, int32_t* labels,
                           void* ws, size_t ws_bytes, void* stream);
/* Smallest core row of every cluster of that same last call (the row that gives a cluster its number in
 * sklearn's sweep): out_rows [nclusters] int32, ascending.  Same workspace rule as the relabel call. */
int pch_dbscan_first_core_rows_i32(int64_t n, int32_t* out_rows, void* ws, size_t ws_bytes, void* stream);

/* Measurement of the radius kernel in the currency that bounds it ("pair-tests/s").  With counting
 * enabled (per thread) pch_dbscan_f32 runs the radius-count kernel in a variant with the SAME control flow that
 * tallies its distance tests; pch_dbscan_pair_stats then reads, from the fit's untouched workspace,
 *   out4_host[0] useful pair tests (a real query point against a real candidate point),
 *   out4_host[1] issued lane slots (64 per wave instruction group: padding and idle lanes included),
 *   out4_host[2] cells that reached the distance-test path, out4_host[3] LDS candidate tiles staged.
 * The counting variant is slower: time the plain kernel, count with this one (bench.py does).  Synchronises. */
void pch_dbscan_set_pair_counting(int enable);
int pch_dbscan_pair_stats(int64_t n, uint64_t* out4_host, void* ws, size_t ws_bytes, void* stream);

/* The (row, cluster) pairs a tile publishes to its neighbour for the cross-tile reconciliation (config 4,
 * pointcloudhookup_amd/tiles.py): of that same last fit, ONE pair per grid cell that holds a core point with
 * x_lo <= x < x_hi - the cell's smallest such row and the cell's cluster id.  All core points of a cell lie within
 * eps of each other, so they share a cluster in any tile that sees them: a pair per cell carries what a pair per
 * point would (a tower on a tile edge: hundreds of pairs instead of hundreds of thousands).  Call it BEFORE
 * pch_dbscan_relabel_i32 (it reports the fit's own ids).  out_pairs [cap,2] int32 in no particular order;
 * *out_count (device int32) = number of such cells, which may exceed cap (then only cap pairs were stored: call
 * again with a larger buffer).  Same workspace rule as the relabel call.
 * Generalises: the per-chunk label offsets of utils/tower_extraction.py:113-116 to tiles that share points. */
int pch_dbscan_strip_pairs_i32(int64_t n, float x_lo, float x_hi, int32_t cap, int32_t* out_pairs,
                               int32_t* out_count, void* ws, size_t ws_bytes, void* stream);

/* One representative per LATTICE cell of up to two strips of a tile: the smallest row among the core points of the
 * cell whose x lies in the strip.  The lattice is shared by all ranks (cell = floor(coordinate / side) per axis, side =
 * eps / sqrt(3) * (1 - 2^-16), anchored at the origin of the common frame), so the two tiles on either side of an edge
 * name the SAME rows for the strip around it - the join on the row links their cluster pieces (tiles.cluster_tiled).
 * xyz [n,3] float32, rows [n] int64 (global row of every point, ascending), labels [n] int32 (-1 = noise), core [n]
 * uint8 - all device; strips_host [nstrips][2] float32 = x_from, x_to per strip (nstrips <= 2, disjoint).
 * out_rows [2][cap] int64, out_labels [2][cap] int32: the pairs of strip k in row k, in no particular order;
 * out_count [4] int32 (device): [k] = cells of strip k (may exceed cap: call again with a larger cap), [2] = flags
 * (1: a coordinate beyond 2^20 cells from the origin, 2: table full - both make the result unusable), [3] = 0.
 * Generalises: the per-chunk label offsets of utils/tower_extraction.py:113-116 to tiles that share points. */
size_t pch_strip_lattice_reps_ws_bytes(int32_t cap);
int pch_strip_lattice_reps_f32(const float* xyz, const int64_t* rows, const int32_t* labels, const uint8_t* core,
                               int64_t n, int32_t nstrips, const float* strips_host, double eps, int32_t cap,
                               int64_t* out_rows, int32_t* out_labels, int32_t* out_count,
                               void* ws, size_t ws_bytes, void* stream);

/* First row of xyz [n,3] float32 that holds NaN or +-inf, -1 if every row is finite.
 * Replaces: sklearn's input validation inside DBSCAN.fit (check_array, ensure_all_finite), which
 * is what makes a chunk "fail" in the reference (utils/tower_extraction.py:107-119).
 * out_row [1] int64 (device). */
int pch_first_nonfinite_row_f32(const float* xyz, int64_t n, int64_t* out_row, void* stream);

/* ------------------------------------------------------------------ stage D0
 * Groups points by cluster label in one pass instead of the reference's K boolean
 * masks.  Replaces: utils/tower_extraction.py:125,131-134.
 * labels [n] int32 in [-1, nclusters)
 * out_perm    [n] int32: point rows ordered by (label, row); noise rows last
 * out_offsets [nclusters+1] int64: cluster k owns out_perm[offsets[k]:offsets[k+1]]
 * out_stats   [nclusters,8] float32 (may be NULL): min xyz, max xyz, 0, 0 per cluster
 */
size_t pch_segment_by_label_ws_bytes(int64_t n, int32_t nclusters);
int pch_segment_by_label(const int32_t* labels, const float* xyz, int64_t n,
                         int32_t nclusters, int32_t* out_perm, int64_t* out_offsets,
                         float* out_stats, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ stage D1, fast mode
 * The reference boxes every cluster with trimesh (utils/tower_extraction.py:137-139:
 * trimesh.PointCloud(cluster_points).bounding_box_oriented): qhull's 3-D hull of the whole cluster, then a
 * search over the hull's facet normals.  These two calls take the bandwidth part and the candidate search
 * off the Python host; the hull itself stays with qhull (pointcloudhookup_amd/obb.py:boxes_fast).
 *
 * pch_obb_shell_f32: for every cluster of the grouped rows (perm / offsets as written by
 * pch_segment_by_label) marks the points that can be vertices of the cluster's convex hull: a point
 * strictly inside a tetrahedron spanned by points of the cluster (support points in 218 fixed directions
 * and their mean) is dropped, everything else - a superset of the hull's vertices, about 1 % - is kept.
 * Clusters of fewer than 2048 points are kept whole.
 * xyz [*,3] float32; perm [n_grouped] int32; offsets [nclusters+1] int64 (device), offsets[nclusters] ==
 * n_grouped; out_keep [n_grouped] uint8 in grouped order. */
size_t pch_obb_shell_ws_bytes(int32_t nclusters);
int pch_obb_shell_f32(const float* xyz, const int32_t* perm, const int64_t* offsets, int32_t nclusters,
                      int64_t n_grouped, uint8_t* out_keep, void* ws, size_t ws_bytes, void* stream);
/* HOST function (no device work): minimum-volume boxes of many convex hulls by trimesh's procedure
 * (bounds.oriented_bounds: facet normals folded to a hemisphere, de-duplicated at 0.1 rad in spherical
 * coordinates, per candidate the minimum-area rectangle of the projected vertices, smallest volume wins),
 * on `nthreads` C++ threads (0: all cores).  All pointers are HOST pointers.
 * verts [sum nv,3] float64 hull vertices, vert_offsets [nhulls+1]; tris [sum nt,3] int32 indices into the
 * hull's own vertices (qhull's simplices), tri_offsets [nhulls+1]; sorted_extents: 0 = [rect_long,
 * rect_short, normal_extent], 1 = ascending with permuted axes (current trimesh).
 * out_to_origin [nhulls,16] row-major 4x4 (world -> box frame), out_extents [nhulls,3],
 * out_status [nhulls]: 0 ok, 1 = no candidate (degenerate hull). */
int pch_obb_min_boxes_f64(const double* verts, const int64_t* vert_offsets, const int32_t* tris,
                          const int64_t* tri_offsets, int32_t nhulls, int32_t sorted_extents,
                          int32_t nthreads, double* out_to_origin, double* out_extents, int32_t* out_status);

/* HOST function: the search alone, for the exact mode.  The caller (pointcloudhookup_amd/obb.py) lets qhull
 * build the hull of the FULL cluster and derives the candidate directions exactly as trimesh does; this call
 * evaluates the box volume of every candidate and names the winner, which the caller then evaluates once
 * more with the reference's own arithmetic - so the result is the python loop's, at 1/50 of its cost.
 * angles [sum nc,2] float64 (theta, phi) in evaluation order, angle_offsets [nhulls+1].
 * out_best [nhulls] int32: index of the first candidate of smallest volume (-1: none);
 * out_volumes [sum nc]: box volume per candidate (inf: no rectangle) - a caller that wants the python loop's
 * winner even when candidates tie in the last bits re-evaluates those that come within its tolerance of the
 * smallest. */
int pch_obb_search_f64(const double* verts, const int64_t* vert_offsets, const double* angles,
                       const int64_t* angle_offsets, int32_t nhulls, int32_t nthreads,
                       int32_t* out_best, double* out_volumes);

/* ------------------------------------------------------------------ viewer helpers
 * Axis-aligned crop, bounds INCLUSIVE, order preserving:
 *   points[(x>=x0)&(x<=x1)&(y>=y0)&(y<=y1)&(z>=z0)&(z<=z1)]
 * Replaces: the per-tower mask + gather of test/kuangxuan.py:69-79 (the box a tower's kuangxuan wire frame shows).
 * xyz [n,3] float64; out_points [n,3] float64 capacity; out_index [n] int64 (may be NULL): source rows;
 * out_count [1] int64.  Rows holding NaN fail the comparisons and are dropped, as in numpy. */
size_t pch_crop_aabb_ws_bytes(int64_t n);
int pch_crop_aabb_f64(const double* xyz, int64_t n, const double* min3_host, const double* max3_host,
                      double* out_points, int64_t* out_index, int64_t* out_count,
                      void* ws, size_t ws_bytes, void* stream);
/* Self-test of the bounded wait: launches eight look-back tiles of which the second never publishes and returns
 * PCH_ERR_TIMEOUT when the six tiles behind it gave up within budget_ms (1..2000) as designed AND the count word they
 * marked (sign bit, as the data-path kernels mark theirs) reads negative; PCH_ERR_HIP when not.  dev_scratch: >= 256 bytes of device memory.  Synchronises.  Not part of the data path (tests only). */
int pch_selftest_lookback_timeout(int budget_ms, void* dev_scratch, size_t scratch_bytes, void* stream);

/* Preview decimation: k distinct rows chosen by a seeded pseudo-random bijection of the row range.
 * Replaces: points[np.random.choice(len(points), k, replace=False)] (pyGUI_towers_test.py:174-177 with k =
 * 200 000, ui/vtk_widget.py:115-118 with k = 500 000).  numpy's draw is unseeded there, so parity is a
 * property: exactly k rows, no row twice, every row from the input, order arbitrary.
 * out_points [k,3] float64 (may be NULL), out_index [k] int64 (may be NULL). */
int pch_decimate_f64(const double* xyz, int64_t n, int64_t k, uint64_t seed,
                     double* out_points, int64_t* out_index, void* stream);

/* ------------------------------------------------------- stages B + C + D0 in one call
 * The body of extract_towers between "points are loaded" and the per-label loop
 * (utils/tower_extraction.py:62-125): pch_ground_filter_f32, pch_dbscan_f32 on the kept points
 * (cell grid sized from the filter's bounding box) and pch_segment_by_label, back to back on
 * `stream` with one shared workspace.  Same results as the three calls.
 * out_points  [n,3] float32, out_index [n] int32 (may be NULL): as pch_ground_filter_f32
 * out_labels  [nf_cap] int32: labels of the kept points (first info->count entries)
 * out_perm    [nf_cap] int32, out_offsets [k_cap+1] int64, out_stats [k_cap,8] float32 (may be
 *             NULL): as pch_segment_by_label; pass out_perm = NULL to skip the grouping
 * nf_cap      upper bound on the kept points the caller is prepared for (<= n); if the filter
 *             keeps more: PCH_ERR_WORKSPACE, info->count says how many (retry with nf_cap >= it)
 * k_cap       capacity of out_offsets / out_stats; more clusters: PCH_ERR_RANGE (labels and
 *             info are valid, group with pch_segment_by_label)
 * info_host   HOST struct, filled on return.  Synchronises (three times: kept count, cell count,
 *             cluster count); the device outputs are complete once `stream` is.
 */
typedef struct PchTowerClusters {
    float   centroid[3];      /* np.mean(raw.astype(f32), axis=0)            (:62-63) */
    float   base;             /* np.percentile(points[:,2], pct)             (:82)    */
    float   threshold;        /* base + offset, or base + fallback_offset    (:87-89) */
    int32_t used_fallback;
    int64_t count_at_offset;  /* points kept by the first threshold */
    float   aabb[6];          /* min xyz, max xyz of the kept (centred) points */
    int64_t count;            /* points kept */
    int32_t nclusters;
    int32_t reserved;
} PchTowerClusters;
size_t pch_tower_clusters_ws_bytes(int64_t n, int64_t nf_cap, int32_t k_cap);
int pch_tower_clusters_f32(const float* raw, int64_t n, double pct, float offset,
                           float fallback_offset, int64_t min_keep, double eps,
                           int32_t min_samples, int64_t chunk_size,
                           float* out_points, int32_t* out_index, int32_t* out_labels,
                           int32_t* out_perm, int64_t* out_offsets, float* out_stats,
                           int64_t nf_cap, int32_t k_cap, PchTowerClusters* info_host,
                           void* ws, size_t ws_bytes, void* stream);

/* --------------------------------------------------------- profiling helpers
 * Last-call timings recorded with hipEvents on the caller's stream when
 * pch_set_profiling(1): fills up to `cap` (name, total ms, launch count) triples for the
 * kernels launched by this thread's pch_* calls since the previous pch_get_profile /
 * pch_set_profiling call.  Returns the number of entries.  (synchronises) */
void pch_set_profiling(int enable);
/* restrict the recording to a comma separated list of kernel names (NULL or "" = all): keeps the
 * host cost of the event records out of a timed region that only needs its heavy kernels */
void pch_set_profiling_filter(const char* names);
int  pch_get_profile(int cap, char names[][48], float* ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* PCH_HIP_H */
