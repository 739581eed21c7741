/* ndt_mi355.h -- C-ABI of the MI355X-native NDT scan-matching core.
 *
 * This is the drop-in boundary for ToySLAM's
 *   pclomp::NormalDistributionsTransform<PointSource, PointTarget>
 * (reference: ndt_omp/include/pclomp/ndt_omp.h:70-502, implementation
 * ndt_omp_impl.hpp, voxel grid voxel_grid_covariance_omp{.h,_impl.hpp}).
 * The header-only adapter include/pclomp/ndt_omp.h re-declares that class on
 * top of these entry points; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C types only; every function returns an ndt_status (0 = ok) unless
 *    noted; ndt_last_error() gives a thread-local message for the last failure.
 *  - point buffers: `n` records of `stride_bytes` bytes, three f32 x,y,z at
 *    offset 0 of each record (pcl::PointXYZ: stride 16; PointXYZI/XYZRGB: 32).
 *  - 4x4 transforms are 16 f32 in COLUMN-major order (Eigen::Matrix4f::data()).
 *  - 6x6 Hessians are 36 f64 row-major; pose vectors are
 *    [tx, ty, tz, roll, pitch, yaw] f64 (ndt_omp_impl.hpp:107-111).
 *  - a handle is thread-compatible (one caller at a time), like the reference
 *    object; distinct handles may be used from distinct threads.
 *  - there is NO CPU fallback: every compute entry point fails with
 *    NDT_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef NDT_MI355_H_
#define NDT_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ndt_context* ndt_handle;

typedef enum {
  NDT_OK = 0,
  NDT_ERR_INVALID = 1,       /* bad argument / call order                          */
  NDT_ERR_NO_DEVICE = 2,     /* no usable HIP device (never falls back to the CPU) */
  NDT_ERR_HIP = 3,           /* HIP runtime error, see ndt_last_error()            */
  NDT_ERR_GRID_OVERFLOW = 4, /* dx*dy*dz > INT32_MAX: voxel_grid_covariance_omp_impl.hpp:75-84 */
  NDT_ERR_NO_INPUT = 5,      /* target/source missing: _impl.hpp:54-60             */
  NDT_ERR_COMM = 6           /* collective callback failed                         */
} ndt_status;

/* pclomp::NeighborSearchMethod, ndt_omp.h:52-57 (same numeric values). */
typedef enum { NDT_KDTREE = 0, NDT_DIRECT26 = 1, NDT_DIRECT7 = 2, NDT_DIRECT1 = 3 } ndt_search_method;

const char* ndt_last_error(void);
/* Number of usable gfx950 devices (0 when none; never an error). */
int ndt_device_count(void);

/* ---- lifetime ----------------------------------------------------------- */
/* NormalDistributionsTransform() ctor, ndt_omp_impl.hpp:47-76: resolution 1.0,
 * step 0.1, outlier 0.55, epsilon 0.1, max_iterations 35, DIRECT7.
 * `device` = HIP device ordinal.  Creating a handle does not touch the GPU
 * until the first target/source upload. */
ndt_status ndt_create(int device, ndt_handle* out);
/* Copy-construction (ndt_omp_mapping_node.cpp:151-169 returns the object by
 * value): parameters and results are copied, the immutable device grid and the
 * source cloud are shared (ref-counted). */
ndt_status ndt_clone(ndt_handle src, ndt_handle* out);
void ndt_destroy(ndt_handle h);

/* ---- parameters (setters mirror ndt_omp.h:115-191 + pcl::Registration) ---- */
ndt_status ndt_set_resolution(ndt_handle h, float resolution);       /* ndt_omp.h:132-142, rebuilds the grid if a source is set */
ndt_status ndt_set_step_size(ndt_handle h, double step_size);        /* :165-169 */
ndt_status ndt_set_outlier_ratio(ndt_handle h, double ratio);        /* :183-187 */
ndt_status ndt_set_transformation_epsilon(ndt_handle h, double eps); /* pcl::Registration */
ndt_status ndt_set_maximum_iterations(ndt_handle h, int n);          /* pcl::Registration */
ndt_status ndt_set_neighborhood_search_method(ndt_handle h, int m);  /* :189-191; unknown values act as DIRECT7 (the reference's `default:`) */
ndt_status ndt_set_num_threads(ndt_handle h, int n);                 /* :115-117; stored, unused on the GPU */
ndt_status ndt_set_min_points_per_voxel(ndt_handle h, int n);        /* voxel_grid_covariance_omp.h:227-239 (clamped to >= 3) */
ndt_status ndt_set_cov_eig_value_inflation_ratio(ndt_handle h, double r); /* .h:253-257 */
float ndt_get_resolution(ndt_handle h);
double ndt_get_step_size(ndt_handle h);
double ndt_get_outlier_ratio(ndt_handle h);

/* ---- inputs --------------------------------------------------------------
 * setInputTarget (ndt_omp.h:122-127): uploads the cloud and builds the voxel
 * grid on the GPU (VoxelGridCovariance::applyFilter, _impl.hpp:48-370).
 * `is_dense` = pcl::PointCloud::is_dense (non-finite points are skipped when 0). */
ndt_status ndt_set_input_target(ndt_handle h, const void* pts, size_t n, size_t stride_bytes, int is_dense);
ndt_status ndt_set_input_source(ndt_handle h, const void* pts, size_t n, size_t stride_bytes);
/* Same, for clouds already resident in HBM (device pointers on the handle's device). */
ndt_status ndt_set_input_target_device(ndt_handle h, const void* d_pts, size_t n, size_t stride_bytes, int is_dense);
ndt_status ndt_set_input_source_device(ndt_handle h, const void* d_pts, size_t n, size_t stride_bytes);
/* Same, BY REFERENCE: the cloud -- dense 16-byte records (x, y, z, anything) on a 16-byte boundary in HBM, pcl::PointXYZ's
 * layout -- is used where it lies instead of being copied: the caller keeps the memory alive and unchanged for as long as
 * it is this handle's input (or the input of a handle cloned from it / sharing it), exactly what pcl::Registration's
 * setInputTarget(ConstPtr) / setInputSource(ConstPtr) promise (ndt_omp.h:122-127: the base class keeps the shared pointer,
 * it never copies the cloud).  Saves the 16 MB -> 16 MB copy of a 1 M-point target; only the bounding boxes are computed. */
ndt_status ndt_set_input_target_device_ref(ndt_handle h, const void* d_pts, size_t n, int is_dense);
ndt_status ndt_set_input_source_device_ref(ndt_handle h, const void* d_pts, size_t n);
/* The voxel index behind the target grid and the prefilter (the reference's std::map<size_t, Leaf> keyed by the linear
 * voxel index, voxel_grid_covariance_omp.h:201): 0 = chosen by occupancy (default), 1 = dense table over the bounding
 * box, 2 = sparse (sort-based build, hash look-up; what a fine leaf over a wide box needs: the 0.1 m prefilter of
 * apps/align.cpp:60-69, kilometre-sized maps).  Results are bit-identical either way. */
ndt_status ndt_set_voxel_index(ndt_handle h, int mode);

/* One uploaded (and spatially ordered) source cloud serving several handles on the same device, the way ndt_clone
 * shares it: `dst` registers the cloud `src` holds.  The levels of a multi-resolution pyramid (BASELINE configs[4]:
 * 2.0 -> 1.0 -> 0.5 m grids over one target, one handle per grid) take every scan from one upload this way, and a
 * second donor handle can upload the next scan on its own stream meanwhile. */
ndt_status ndt_share_input_source(ndt_handle dst, ndt_handle src);
/* The same for the target: `dst` takes the target cloud AND the voxel grid `src` built from it (no copy, no rebuild;
 * resolution, min_points_per_voxel and the eigenvalue ratio of `dst` become the grid's). */
ndt_status ndt_share_input_target(ndt_handle dst, ndt_handle src);

/* CU partitions: overlapping the NEXT scan's preparation with the CURRENT registration.
 * A registration keeps the whole chip busy for its duration -- the persistent evaluation kernel holds one workgroup per CU,
 * the per-evaluation kernels of a multi-million-point scan fill every CU -- so the upload, spatial ordering and target grid
 * build of the following scan, issued meanwhile by another handle on another stream, wait behind it (the reference's nodes
 * do the two strictly one after the other: ndt_omp_mapping_node.cpp:151-169, 195-211).  With partitions the two never
 * compete: the handle's stream is created with a CU mask (hipExtStreamCreateWithCUMask),
 *   0  the whole device (default);
 *   1  the registration partition: all CUs but the last NDT_SIDE_CUS (default 32) -- for the handles that align;
 *   2  the side partition: those NDT_SIDE_CUS CUs -- for handles that only prepare inputs (ndt_set_input_*,
 *      ndt_voxel_grid_filter*, ndt_map_update*) and hand them over with ndt_share_input_source / ndt_share_input_target.
 * The latency kernels cut a scan into one workgroup per CU of the handle's partition, so a partitioned handle adds the same
 * per-point terms in another (still fixed) order: sums equal to the unpartitioned handle's to the last bits of an f64, as
 * between the lock-step batches and ndt_align; grids, voxel filters and map updates are bit-identical in any partition.
 * Call before the handle is used, or between calls (the handle waits for its work in flight and moves to the stream of the new partition; it keeps the streams it has had until ndt_destroy).
 * ndt_get_cu_partition reports the partition and the CUs the stream really got (the device's count when masks are
 * unavailable). */
ndt_status ndt_set_cu_partition(ndt_handle h, int partition);
ndt_status ndt_get_cu_partition(ndt_handle h, int* partition, int* n_cus);

/* ---- registration --------------------------------------------------------
 * pcl::Registration::align(output, guess) -> computeTransformation
 * (ndt_omp_impl.hpp:80-171) with the More-Thuente line search (:772-932).
 * guess == NULL means Identity.  out_cloud (optional, host memory) receives the source
 * transformed by the last line-search trial, n_source records of
 * out_stride_bytes (x,y,z,1.0f written at offset 0).  With out_cloud == NULL the call
 * returns as soon as the result is known; the aligned cloud is then completed in stream
 * order and ndt_get_output_device waits for it. */
ndt_status ndt_align(ndt_handle h, const float* guess, float* final_transformation, int* has_converged,
                     int* final_num_iteration, double* transformation_probability, void* out_cloud,
                     size_t out_stride_bytes);
/* Results of the last align (hasConverged / getFinalTransformation /
 * getFinalNumIteration / getTransformationProbability). */
ndt_status ndt_get_result(ndt_handle h, float* final_transformation, int* has_converged, int* final_num_iteration,
                          double* transformation_probability);
/* Device pointer (n_source x float4) of the last align's transformed source. */
ndt_status ndt_get_output_device(ndt_handle h, const void** d_cloud, size_t* n);
/* Work counters of the last align: derivative evaluations E, f64 Hessian
 * recomputes, mean valid neighbours per point (h-bar) of the last evaluation.
 * After ndt_align_batch: the scan evaluations / f64 recomputes of all members
 * together and h-bar over all of their evaluations. */
ndt_status ndt_get_stats(ndt_handle h, int* n_evals, int* n_hessian_recomputes, double* mean_neighbors);

/* calculateScore(cloud), ndt_omp_impl.hpp:935-983 (cloud is used as given). */
ndt_status ndt_calculate_score(ndt_handle h, const void* cloud, size_t n, size_t stride_bytes, double* score);

/* [PCL 1.10] pcl::Registration::getFitnessScore(max_range) (printed by apps/align.cpp:24-33 and
 * ndt_rosbag_mapping_node.cpp:133): the source transformed by the last align's final transformation,
 * nearest target point of each (exact search, f32 squared distances as FLANN's L2_Simple), mean of the
 * squared distances that are <= max_range (PCL compares the SQUARED distance with max_range; kept),
 * DBL_MAX when none qualifies.  Runs on the GPU over the target's voxel grid; pass DBL_MAX for PCL's
 * default.  Non-finite source points are skipped. */
ndt_status ndt_get_fitness_score(ndt_handle h, double max_range, double* fitness);

/* ---- scan prefilter (row N1 of the scope table) -----------------------------
 * pcl::VoxelGrid<PointT>::filter -- one centroid per occupied voxel, output in ascending
 * voxel-index order -- as every caller runs it before NDT (ndt_omp/apps/align.cpp:60-69,
 * ndt_omp_mapping_node.cpp:142-148,203-210; [PCL 1.10] filters/impl/voxel_grid.hpp).  xyz only.
 * `out` must hold n records of out_stride_bytes (x,y,z,1.0f at offset 0); *n_out = voxels written.
 * When the voxel index space would overflow int32 PCL warns and passes the input through: the
 * input is then copied to `out`, *n_out = n and NDT_ERR_GRID_OVERFLOW is returned. */
ndt_status ndt_voxel_grid_filter(ndt_handle h, const void* pts, size_t n, size_t stride_bytes, int is_dense, float leaf_size,
                                 void* out, size_t out_stride_bytes, size_t* n_out);
/* Same with input and output (n x float4) resident in HBM. */
ndt_status ndt_voxel_grid_filter_device(ndt_handle h, const void* d_pts, size_t n, size_t stride_bytes, int is_dense,
                                        float leaf_size, void* d_out_float4, size_t* n_out);

/* ---- global map accumulation (row N2 of the scope table) ---------------------
 * update_global_map of the mapping nodes (ndt_omp_mapping_node.cpp:195-211,
 * ndt_rosbag_mapping_node.cpp:146-161): pcl::transformPointCloud(scan, pose) -> global_map += it ->
 * global_map = VoxelGrid(leaf).filter(global_map).  The map lives in HBM with the handle; `pose` is a
 * column-major 4x4 (NULL = identity).  *overflowed = 1 when the leaf is too small for the map's
 * bounding box: PCL then keeps the unfiltered concatenation, and so does this.
 * ndt_host_chain_pose is the nodes' `pose = pose * transform` (:88-99 / :62-68) in Eigen's f32 rounding. */
ndt_status ndt_map_clear(ndt_handle h);
ndt_status ndt_map_update(ndt_handle h, const void* scan, size_t n, size_t stride_bytes, int is_dense, const float* pose,
                          float leaf_size, int* overflowed);
ndt_status ndt_map_update_device(ndt_handle h, const void* d_scan, size_t n, size_t stride_bytes, int is_dense,
                                 const float* pose, float leaf_size, int* overflowed);
ndt_status ndt_map_size(ndt_handle h, size_t* n);
ndt_status ndt_map_get(ndt_handle h, void* out, size_t out_stride_bytes); /* x,y,z,1.0f per point */
ndt_status ndt_map_get_device(ndt_handle h, const void** d_pts_float4, size_t* n);
void ndt_host_chain_pose(const float* pose /*16*/, const float* transform /*16*/, float* out /*16, may alias*/);

/* What a node does once, at start-up, instead of inside its first scans: the device context, the library's code object (tens
 * of milliseconds on first use), the handle's page-locked result / staging slots, and one pass through the loop's calls
 * (voxel filter, target grid, registration, map update) on a synthetic scan of `expected_scan_points` points, so that every
 * kernel has been launched once and the handle's memory pool holds blocks of the sizes the real scans will ask for.
 * The handle's inputs, last result and map are left as they were. */
ndt_status ndt_warm_up(ndt_handle h, size_t expected_scan_points);

/* ---- clouds that stay in HBM: the node loop without host round trips -----------------------------------------------
 * In all three mapping nodes a filtered scan is used four times: as the output of the prefilter
 * (ndt_omp_mapping_node.cpp:142-148), as the source of the registration against its predecessor (:151-169), as the target of
 * the registration of its successor (cloud k is clouds_[current_index_] of one pair and clouds_[current_index_ - 1] of the
 * next, :77-79), and as what update_global_map adds to the map (:195-211).  Through host buffers that is one download and
 * three uploads, repacks and bounding-box passes of the same points.  An `ndt_cloud` is that scan as an object: dense
 * 16-byte records in HBM together with their bounding boxes, reference-counted (a handle that takes it as an input holds a
 * reference of its own, so releasing the caller's reference is always safe).
 *   ndt_cloud_voxel_filter      N1 with the result left in HBM (pts: host memory, or device memory when on_device != 0)
 *   ndt_cloud_upload            a host cloud as it is
 *   ndt_set_input_source_cloud / ndt_set_input_target_cloud / ndt_map_update_cloud
 *                               the three consumers, by reference: no copy, no repack, no bounding-box pass
 *   ndt_promote_source_to_target  the handle's current input source becomes its input target (cloud k of the pair
 *                               (k-1, k) is the target of the pair (k, k+1)); the voxel grid is built from the
 *                               resident points -- for callers that kept no ndt_cloud, e.g. after ndt_set_input_source
 * Results are bit-identical to the host-buffer entry points (same kernels on the same points).  A cloud belongs to the
 * device of the handle that made it; handles on other streams of that device may use it (the library orders the streams). */
typedef struct ndt_cloud_s* ndt_cloud;
ndt_status ndt_cloud_voxel_filter(ndt_handle h, const void* pts, size_t n, size_t stride_bytes, int is_dense, float leaf_size,
                                  int on_device, ndt_cloud* out, int* overflowed);
/* N1 of an ndt_cloud in two halves: _begin queues the whole filter on a stream of the handle's own and returns at once,
 * _end waits for it and hands out the result -- in between the caller registers the PREVIOUS scan (the prefilter of scan
 * k + 1 runs beside the registration of scan k).  One prefilter at a time per handle.  Same result as ndt_cloud_voxel_filter. */
ndt_status ndt_cloud_voxel_filter_begin(ndt_handle h, ndt_cloud in, int is_dense, float leaf_size);
ndt_status ndt_cloud_voxel_filter_end(ndt_handle h, ndt_cloud* out, int* overflowed);
ndt_status ndt_cloud_upload(ndt_handle h, const void* pts, size_t n, size_t stride_bytes, ndt_cloud* out);
ndt_status ndt_cloud_size(ndt_cloud c, size_t* n);
ndt_status ndt_cloud_data(ndt_cloud c, const void** d_pts_float4, size_t* n);               /* the records in HBM */
ndt_status ndt_cloud_download(ndt_handle h, ndt_cloud c, void* out, size_t out_stride_bytes); /* x,y,z,1.0f per point */
void ndt_cloud_release(ndt_cloud c);
ndt_status ndt_set_input_source_cloud(ndt_handle h, ndt_cloud c);
ndt_status ndt_set_input_target_cloud(ndt_handle h, ndt_cloud c, int is_dense);
ndt_status ndt_map_update_cloud(ndt_handle h, ndt_cloud scan, int is_dense, const float* pose, float leaf_size, int* overflowed);
ndt_status ndt_promote_source_to_target(ndt_handle h, int is_dense);

/* ---- PCD files (row N3 of the scope table) -----------------------------------
 * What pcl::io::loadPCDFile<pcl::PointXYZ> hands the callers (ndt_omp/apps/align.cpp:48-55,
 * ndt_omp_mapping_node.cpp:140, ndt_omp_node.cpp:82) and what pcl::io::savePCDFileBinary writes
 * (lidar_subscriber_node.cpp:46): PCD v0.7, DATA ascii | binary | binary_compressed; x, y, z are
 * picked by field name, other fields (intensity, rgb, ...) are skipped.  Host only, no device needed.
 * data_kind: 0 ascii, 1 binary, 2 binary_compressed.  Records written to `out` are x,y,z(,1.0f when
 * stride_bytes >= 16); *is_dense = every point finite (PCDReader's rule). */
ndt_status ndt_pcd_read_header(const char* path, size_t* n_points, int* n_fields, int* data_kind);
ndt_status ndt_pcd_read_xyz(const char* path, void* out, size_t capacity_points, size_t stride_bytes, size_t* n_points,
                            int* is_dense);
ndt_status ndt_pcd_write_xyz(const char* path, const void* pts, size_t n, size_t stride_bytes, int binary);

/* A directory of numbered scans, as the mapping node consumes it (lidar_subscriber/src/ndt_omp_mapping_node.cpp):
 * process_new_clouds (:110-136) lists the *.pcd files whose number -- the integer after the last '_' of the file
 * stem (extract_file_number, :231-239) -- is >= loaded_clouds + 1 and loads them in ascending order; the node calls
 * it once at start-up and then once per second (:28-34, directory polling).  ndt_pcd_sequence_poll is that listing;
 * ndt_pcd_sequence_next hands out the queued files one by one as x, y, z, 1.0f records (16 bytes) in page-locked host
 * memory, and while the caller works on one scan the next file is read and parsed by a background thread (the two
 * buffers alternate; a scan stays valid until the following call of ndt_pcd_sequence_next).
 * next: *pts == NULL when nothing is queued.  A file that cannot be parsed returns NDT_ERR_INVALID and is skipped,
 * like load_and_filter_cloud's nullptr (:138-141).  Works without a device (pageable buffers then). */
typedef struct ndt_pcd_sequence* ndt_pcd_sequence_handle;
ndt_status ndt_pcd_sequence_open(const char* directory, ndt_pcd_sequence_handle* out);
ndt_status ndt_pcd_sequence_poll(ndt_pcd_sequence_handle s, size_t loaded_clouds, size_t* n_new_files);
ndt_status ndt_pcd_sequence_next(ndt_pcd_sequence_handle s, const void** pts, size_t* n, int* is_dense, int* file_number);
/* Scans staged into HBM by the reader: after ndt_pcd_sequence_stage(s, device) every scan is copied to the device as soon as
 * its file has been read (by the reading thread, on a copy stream of the sequence's own), and ndt_pcd_sequence_next_device
 * hands out the device records (16 bytes each, valid until the next call) together with the host ones -- the upload of scan
 * k + 1 runs while the caller registers scan k.  *d_pts == NULL with NDT_OK: nothing queued, as _next. */
ndt_status ndt_pcd_sequence_stage(ndt_pcd_sequence_handle s, int device);
ndt_status ndt_pcd_sequence_next_device(ndt_pcd_sequence_handle s, const void** d_pts, const void** host_pts, size_t* n, int* is_dense,
                                        int* file_number);
/* The same scan as an ndt_cloud: a VIEW of the staged records (the sequence owns the memory: valid until the next call of
 * a _next* function) that carries the bounding boxes the reading thread computed on the host while the copy ran -- a
 * prefilter of it needs nothing from the device before its kernels can be queued.  *cloud == NULL with NDT_OK: nothing
 * queued.  Release the view with ndt_cloud_release. */
ndt_status ndt_pcd_sequence_next_cloud(ndt_pcd_sequence_handle s, ndt_cloud* cloud, const void** host_pts, size_t* n, int* is_dense,
                                       int* file_number);
void ndt_pcd_sequence_close(ndt_pcd_sequence_handle s);
/* extract_file_number (:231-239) */
int ndt_host_extract_file_number(const char* file_stem);
/* pcl::fromROSMsg(sensor_msgs::PointCloud2, PointCloud<PointXYZ>) for the layouts lidar drivers publish
 * (ndt_rosbag_mapping_node.cpp:45-50): n records of point_step bytes with three f32 fields at byte offsets
 * off_x, off_y, off_z -- any step and any offsets, aligned or not (e.g. the 22-byte x,y,z,intensity,ring,time
 * records of a Velodyne driver) -- repacked into x, y, z, 1.0f records of 16 bytes, which is what every other
 * entry point takes with stride 16.  Host only.  *is_dense = every coordinate finite. */
ndt_status ndt_host_repack_fields(const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                                  void* out_xyz1, int* is_dense);

/* ---- batch (map-build mode: many sources against the one target) ----------
 * Registers n_scans sources in lock-step, one fused derivative launch per
 * line-search step for the whole batch.  Scan k is points
 * [offsets[k], offsets[k+1]) of `pts`.  guesses == NULL -> Identity.
 * Per-scan outputs are arrays of n_scans entries (any may be NULL). */
ndt_status ndt_align_batch(ndt_handle h, const void* pts, const size_t* offsets /* n_scans+1 */, size_t n_scans,
                           size_t stride_bytes, const float* guesses /* n_scans*16 or NULL */,
                           float* final_transformations /* n_scans*16 */, int* has_converged,
                           int* final_num_iteration, double* transformation_probability);
ndt_status ndt_align_batch_device(ndt_handle h, const void* d_pts, const size_t* offsets, size_t n_scans,
                                  size_t stride_bytes, const float* guesses, float* final_transformations,
                                  int* has_converged, int* final_num_iteration, double* transformation_probability);

/* How many independent lock-step groups ndt_align_batch* runs the batch as (each on a stream and a host thread of
 * its own, so one group's host-side Newton / More-Thuente steps and launches hide behind the other groups' kernels):
 * 0 = automatic (2 from 16 scans, 4 from 192), 1 = one lock-step loop.  Every scan's result is independent of the
 * members of its group and of the grouping: every scan is ordered on a lattice of its own and summed in its own blocks, so
 * a scan gets the same bits in any batch, group or rank.  Batches with an exchange step (communicator / all-reduce hook)
 * always run as one loop. */
ndt_status ndt_set_batch_groups(ndt_handle h, int n_groups);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ---------------------------
 * The reference is a single process (ndt_omp_impl.hpp:206 is its only parallel construct); this is the exchange step
 * north_star adds.  Registrations of different scans are independent, so ndt_align_batch* on every rank over its own
 * share of the scans needs NO collective (what bench.py measures at N > 1).  Two uses of a communicator remain:
 *  - ndt_align_batch_sharded*: the literal lock-step form -- the batch has total_scans scans, this rank holds scans
 *    [first_scan, first_scan + n_local) (offsets: n_local + 1 entries into ITS points); every rank steps all
 *    total_scans Newton / More-Thuente state machines, rows of the scans a rank does not hold are zero, and ONE in-place
 *    SUM all-reduce of the [total_scans][NDT_EVAL_STRIDE] f64 buffer per lock-step gives every rank every row.
 *    guesses and the per-scan outputs have total_scans entries and come out identical on every rank.
 *  - ndt_align with a communicator set: the source cloud each rank holds is a SHARD of one big scan (target replicated);
 *    the 32-f64 row of every evaluation is all-reduced, so all ranks walk the same registration.
 * PRECONDITION of both: every rank holds the SAME target (same points, same parameters) -- the grid is replicated, not
 * exchanged.  A rank whose target has no voxel at all still joins the exchange of a sharded batch with zero rows; in the
 * point-sharded ndt_align such a rank returns without a collective, which is consistent only when every rank's target
 * is that empty one.  If a peer never joins a collective the waiting rank gives up after NDT_BATCH_TIMEOUT_S (60)
 * seconds, aborts its communicator (ncclCommAbort, so that its stream drains) and returns NDT_ERR_COMM.
 * The collective is issued from C++ on the handle's own stream (no host synchronisation around it); librccl is loaded
 * on first use.  ndt_comm_get_unique_id (rank 0) wraps ncclGetUniqueId; the caller carries the NDT_COMM_ID_BYTES to
 * the other ranks (MPI, a file, torch.distributed ...), then every rank calls ndt_comm_init_rank on its own device. */
#define NDT_COMM_ID_BYTES 128
ndt_status ndt_comm_get_unique_id(void* id_out /* NDT_COMM_ID_BYTES */);
ndt_status ndt_comm_init_rank(ndt_handle h, const void* id, int rank, int world_size);
ndt_status ndt_comm_destroy(ndt_handle h);
/* rank / world size of the handle's communicator (-1 / 0 without one), collectives issued since ndt_comm_init_rank,
 * lock-steps of the last ndt_align_batch* call (any may be NULL) */
ndt_status ndt_comm_stats(ndt_handle h, int* rank, int* world_size, long long* n_collectives, int* lock_steps);
ndt_status ndt_align_batch_sharded(ndt_handle h, const void* pts, const size_t* offsets /* n_local+1 */, size_t n_local,
                                   size_t first_scan, size_t total_scans, size_t stride_bytes,
                                   const float* guesses /* total_scans*16 or NULL */, float* final_transformations /* total_scans*16 */,
                                   int* has_converged, int* final_num_iteration, double* transformation_probability);
ndt_status ndt_align_batch_sharded_device(ndt_handle h, const void* d_pts, const size_t* offsets, size_t n_local,
                                          size_t first_scan, size_t total_scans, size_t stride_bytes, const float* guesses,
                                          float* final_transformations, int* has_converged, int* final_num_iteration,
                                          double* transformation_probability);

/* Caller-supplied exchange step (tests over gloo; any collective library the caller already has):
 * after every fused evaluation the packed [n_rows][NDT_EVAL_STRIDE] f64 result
 * buffer is handed to `fn` for an in-place SUM all-reduce across ranks
 * (the device buffer when `on_device` != 0, a host copy otherwise).  Not stream-ordered: the library synchronises
 * around the call.  A communicator set with ndt_comm_init_rank takes precedence.  fn returns 0 on success. */
#define NDT_EVAL_STRIDE 32 /* score, g[6], H upper-tri[21], n_neighbors, 3 spare */
typedef int (*ndt_allreduce_fn)(void* buf, size_t n_doubles, int on_device, void* user);
ndt_status ndt_set_allreduce(ndt_handle h, ndt_allreduce_fn fn, void* user, int on_device);

/* ---- inspection / test entry points --------------------------------------
 * One computeDerivatives evaluation (ndt_omp_impl.hpp:179-285) at pose p; the
 * source is transformed by T(p) as computeStepLengthMT does (:827-837).
 * H may be NULL (compute_hessian = false). */
ndt_status ndt_eval(ndt_handle h, const double* p, double* score, double* gradient, double* hessian,
                    double* mean_neighbors);
/* Same with an explicit 4x4 (column-major) applied to the source while the
 * angle derivatives come from p -- the initial evaluation of align with a
 * non-identity guess (:95-119). */
ndt_status ndt_eval_with_matrix(ndt_handle h, const float* T, const double* p, double* score, double* gradient,
                                double* hessian, double* mean_neighbors);
/* computeHessian (:540-645): the all-f64 Hessian at pose p. */
ndt_status ndt_eval_hessian_f64(ndt_handle h, const double* p, double* hessian);

/* Target grid (VoxelGridCovariance::leaves_): number of occupied voxels and a
 * dump in ascending linear-index order.  cov/icov are row-major 3x3 f64;
 * nr_points is -1 for rejected voxels as in _impl.hpp:337-341,360-364. */
ndt_status ndt_grid_size(ndt_handle h, size_t* n_leaves, size_t* n_valid);
ndt_status ndt_grid_info(ndt_handle h, int* min_b /*3*/, int* max_b /*3*/, int* div_b /*3*/);
ndt_status ndt_grid_dump(ndt_handle h, int64_t* idx, int* nr_points, double* mean, double* cov, double* icov,
                         double* evals);

/* Live kernel timing with HIP events recorded on the handle's own stream (bench.py's roofline leg).
 * on = 1: ndt_align runs one launch per evaluation and brackets each with an event pair; kind 0 =
 *         derivatives with Hessian, 1 = without, 2 = f64 Hessian.  ndt_align_batch brackets the
 *         derivative kernels of every lock-step with one pair: kind 0, n_launches = lock-steps.
 *         With a communicator set (ndt_align_batch_sharded) every lock-step is split three ways: kind 0 the
 *         derivative kernels, kind 4 k_reduce + ncclAllReduce, kind 5 k_publish_rows (all on the library stream),
 *         and kind 6 the host's wall time for the whole lock-step (descriptors, launches, wait, solver steps), in ms.
 * on = 2: ndt_align keeps its persistent kernel (one launch per registration, the kernel of the
 *         timed region) and brackets that launch with one event pair; kind 3.
 * Off (0) by default: the event records cost host time. */
ndt_status ndt_profile_enable(ndt_handle h, int on);

/* How ndt_align evaluates: 1 (default) = one persistent kernel per registration, fed one command
 * per evaluation through a pinned mailbox; 0 = one kernel launch per evaluation.  Both run the same
 * device code over the same thread partition and return bit-identical results (tests pin that);
 * the setting only trades latency.  The NDT_PERSISTENT environment variable sets the default.
 * Unless this call has insisted on 1, a registration that starts while another ndt_align of this process is
 * in flight on the same device uses the launch path: the persistent kernel holds its CUs for a whole registration, so
 * concurrent callers would otherwise take turns at it; calling this with 1 insists on the persistent kernel. */
ndt_status ndt_set_evaluation_path(ndt_handle h, int persistent);
ndt_status ndt_profile_read(ndt_handle h, int kind, long long* n_launches, double* total_ms, int reset);

/* Device self-test of the wave64 fold reduction used by every kernel epilogue: n_blocks blocks
 * of known per-thread values; block_sums receives n_blocks x NDT_EVAL_STRIDE doubles (slots 0..28). */
ndt_status ndt_selftest_reduce(ndt_handle h, int n_blocks, double* block_sums);
/* liveness of the persistent evaluation server: one evaluation, a host stall of stall_ms (the server's patience is
 * 20 ms), one more request (*served_after_stall = 0 when the server had left, as it must for stall_ms > 20), then the
 * same evaluation through the launch path and through a fresh server; scores[3] = the three results. */
ndt_status ndt_selftest_server_idle(ndt_handle h, const double* p, int stall_ms, int* served_after_stall, double* scores);

/* Diagnostic (development aid): one DIRECT7 evaluation at pose p by the s_memtime-stamped build of
 * the derivative kernel.  stamps receives n_waves x 8 u64 (shader cycles at: entry, point
 * arrived, LUT arrived, first record arrived, neighbour math done, wave fold done, block done);
 * *n_waves in: capacity, out: waves written. */
ndt_status ndt_diag_stamps(ndt_handle h, const double* p, unsigned long long* stamps, size_t* n_waves);

/* Diagnostic: round-trip latency of the persistent evaluation server, averaged over n_iter commands:
 * us[0] = no-op round (protocol only), us[1] = derivatives without Hessian, us[2] = with Hessian. */
ndt_status ndt_diag_server_roundtrip(ndt_handle h, const double* p, int n_iter, double* us);
/* Diagnostic: `rounds` evaluations at pose p driven from the DEVICE (the last arriving block adds the part sums and posts the
 * next command itself; no Newton / line-search step): us[0] per round without the per-point body, us[1] with the
 * with-Hessian body -- the floor a device-side solver would start from.  DIRECT7 only. */
ndt_status ndt_diag_selfdrive(ndt_handle h, const double* p, int rounds, double* us);

/* Host-side scalar pieces of the driver (no GPU needed), exported so that the
 * CPU test-suite can check them against the oracle. */
void ndt_host_solve6(const double* H /*36 row-major*/, const double* b /*6*/, double* x /*6*/); /* JacobiSVD.solve, :127-129 */
void ndt_host_pose_to_matrix(const double* p /*6*/, float* T /*16 col-major*/);                  /* :146-149, 827-830 */
void ndt_host_matrix_to_pose(const float* T /*16 col-major*/, double* p /*6*/);                  /* :103-111 */
void ndt_host_angle_derivatives(const double* p /*6*/, float* j_ang /*8*3*/, float* h_ang /*15*3*/,
                                double* j_ang_d /*8*3*/, double* h_ang_d /*15*3*/);              /* :288-395 */
void ndt_host_gauss(float resolution, double outlier_ratio, double* d /*3: d1,d2,d3*/);          /* :86-93 */
/* Host threads a lock-step batch may use.  ndt_host_thread_budget probes this process: CPUs in its affinity mask, the
 * cgroup's CPU bandwidth in CPUs (cpu.max / cfs_quota; 0 = unlimited) and LOCAL_WORLD_SIZE (ranks sharing the node, 1
 * when unset).  ndt_host_thread_plan is the pure rule applied to such a budget: share = min(affinity, floor(quota)) /
 * local_world_size (at least 1); *pool_threads = clamp(share / 2, 1, 16) workers for the per-step solver work (they
 * block when idle), *max_batch_groups = clamp(share, 1, 8) independent lock-step groups (a host thread each).
 * NDT_HOST_THREADS / NDT_BATCH_GROUPS / ndt_set_batch_groups override. */
void ndt_host_thread_budget(int* affinity_cpus, double* quota_cpus, int* local_world_size);
void ndt_host_thread_plan(int affinity_cpus, double quota_cpus, int local_world_size, int* pool_threads, int* max_batch_groups);
/* Runs the Newton + More-Thuente driver against a caller-supplied evaluator
 * (test hook: lets the CPU suite drive the PRODUCT driver with oracle
 * evaluations).  kind: 0 = derivatives with Hessian, 1 = without, 2 = f64
 * Hessian only.  T is the 4x4 (col-major) to apply to the source. */
typedef int (*ndt_eval_cb)(void* user, int kind, const float* T, const double* p, double* score, double* g, double* H);
ndt_status ndt_host_run_driver(ndt_eval_cb cb, void* user, size_t n_source, const float* guess, float resolution,
                               double step_size, double outlier_ratio, double trans_eps, int max_iter,
                               float* final_transformation, int* has_converged, int* final_num_iteration,
                               double* transformation_probability, int* n_evals, int* n_hessian_recomputes);

#ifdef __cplusplus
}
#endif
#endif /* NDT_MI355_H_ */
