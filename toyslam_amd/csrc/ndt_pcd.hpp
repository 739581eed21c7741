// ndt_pcd.hpp -- PCD v0.7 reader / writer (host, no PCL); see ndt_pcd.cpp.
#pragma once
#include <cstddef>
#include <string>

namespace ndt {
// all return 0 on success; 1 = cannot open, 2 = malformed file, 3 = output buffer too small
int pcd_read_header(const char* path, size_t* n_points, int* n_fields, int* data_kind, std::string& err);
int pcd_read_xyz(const char* path, void* out, size_t capacity_points, size_t stride_bytes, size_t* n_points, int* is_dense,
                 std::string& err);
int pcd_write_xyz(const char* path, const void* pts, size_t n, size_t stride_bytes, int binary, std::string& err);
}  // namespace ndt
