// ndt_pcd.cpp -- PCD v0.7 files either side of the registration path (row N3 of the scope table):
// what pcl::io::loadPCDFile<pcl::PointXYZ> hands the callers (ndt_omp/apps/align.cpp:48-55,
// ndt_omp_mapping_node.cpp:140, ndt_omp_node.cpp:82) and what pcl::io::savePCDFileBinary writes
// (lidar_subscriber_node.cpp:46).  Host code, no PCL: header parser, ascii / binary /
// binary_compressed (LZF, struct-of-arrays) bodies, x y z picked by field name.
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ndt_pcd.hpp"

#include <emmintrin.h>
#include <sys/mman.h>
#include <unistd.h>

namespace ndt {
namespace {

struct Field {
  std::string name;
  int size = 4;
  char type = 'F';
  int count = 1;
  size_t offset = 0;  // byte offset inside one point record
};

struct Header {
  std::vector<Field> fields;
  size_t width = 0, height = 1, points = 0;
  bool have_points = false;
  int data = -1;  // 0 ascii, 1 binary, 2 binary_compressed
  size_t record = 0;
  long body_offset = 0;
  size_t body_bytes = 0;  // bytes of the file after the header
};

std::vector<std::string> split(const std::string& s) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r')) i++;
    size_t j = i;
    while (j < s.size() && s[j] != ' ' && s[j] != '\t' && s[j] != '\r') j++;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
  return out;
}

bool read_line(FILE* f, std::string& line) {
  line.clear();
  int c;
  while ((c = std::fgetc(f)) != EOF) {
    if (c == '\n') return true;
    line.push_back(static_cast<char>(c));
  }
  return !line.empty();
}

std::string upper(std::string s) {
  for (char& c : s)
    if (c >= 'a' && c <= 'z') c = static_cast<char>(c - 'a' + 'A');
  return s;
}

// returns an empty string on success, else what is wrong
std::string parse_header(FILE* f, Header& h) {
  std::string line;
  bool got_fields = false;
  while (read_line(f, line)) {
    std::vector<std::string> tok = split(line);
    if (tok.empty() || tok[0][0] == '#') continue;
    const std::string key = upper(tok[0]);
    if (key == "VERSION") continue;
    if (key == "FIELDS" || key == "COLUMNS") {
      h.fields.clear();
      for (size_t i = 1; i < tok.size(); i++) {
        Field fd;
        fd.name = tok[i];
        h.fields.push_back(fd);
      }
      got_fields = true;
    } else if (key == "SIZE") {
      if (tok.size() - 1 != h.fields.size()) return "SIZE does not match FIELDS";
      for (size_t i = 1; i < tok.size(); i++) h.fields[i - 1].size = std::atoi(tok[i].c_str());
    } else if (key == "TYPE") {
      if (tok.size() - 1 != h.fields.size()) return "TYPE does not match FIELDS";
      for (size_t i = 1; i < tok.size(); i++) h.fields[i - 1].type = static_cast<char>(upper(tok[i])[0]);
    } else if (key == "COUNT") {
      if (tok.size() - 1 != h.fields.size()) return "COUNT does not match FIELDS";
      for (size_t i = 1; i < tok.size(); i++) h.fields[i - 1].count = std::atoi(tok[i].c_str());
    } else if (key == "WIDTH" && tok.size() > 1) {
      h.width = std::strtoull(tok[1].c_str(), nullptr, 10);
    } else if (key == "HEIGHT" && tok.size() > 1) {
      h.height = std::strtoull(tok[1].c_str(), nullptr, 10);
    } else if (key == "VIEWPOINT") {
      continue;
    } else if (key == "POINTS" && tok.size() > 1) {
      h.points = std::strtoull(tok[1].c_str(), nullptr, 10);
      h.have_points = true;
    } else if (key == "DATA" && tok.size() > 1) {
      const std::string kind = upper(tok[1]);
      h.data = kind == "ASCII" ? 0 : kind == "BINARY" ? 1 : kind == "BINARY_COMPRESSED" ? 2 : -1;
      if (h.data < 0) return "unknown DATA kind";
      break;
    }
  }
  if (!got_fields) return "no FIELDS line";
  if (h.data < 0) return "no DATA line";
  if (!h.have_points) h.points = h.width * h.height;
  size_t off = 0;
  for (Field& fd : h.fields) {
    if (fd.size <= 0 || fd.size > 8 || fd.count < 0 || fd.count > (1 << 20)) return "bad SIZE / COUNT";
    fd.offset = off;
    off += static_cast<size_t>(fd.size) * fd.count;
  }
  if (off == 0 || off > (size_t(1) << 24)) return "bad record size";
  h.record = off;
  h.body_offset = std::ftell(f);
  // what is left of the file bounds what the header may promise (no allocation on a lying header)
  if (std::fseek(f, 0, SEEK_END) != 0) return "cannot seek";
  h.body_bytes = static_cast<size_t>(std::ftell(f) - h.body_offset);
  if (std::fseek(f, h.body_offset, SEEK_SET) != 0) return "cannot seek";
  if (h.data == 1 && h.points > h.body_bytes / h.record) return "truncated binary body";
  if (h.data == 0 && h.points > h.body_bytes) return "fewer ascii records than POINTS";
  if (h.data == 2 && h.points > (size_t(1) << 40) / h.record) return "implausible POINTS";
  return "";
}

// one scalar of a binary record -> float
float scalar_to_float(const unsigned char* p, const Field& fd) {
  switch (fd.type) {
    case 'F':
      if (fd.size == 4) { float v; std::memcpy(&v, p, 4); return v; }
      if (fd.size == 8) { double v; std::memcpy(&v, p, 8); return static_cast<float>(v); }
      break;
    case 'U':
      if (fd.size == 1) return static_cast<float>(*p);
      if (fd.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return static_cast<float>(v); }
      if (fd.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return static_cast<float>(v); }
      if (fd.size == 8) { uint64_t v; std::memcpy(&v, p, 8); return static_cast<float>(v); }
      break;
    case 'I':
      if (fd.size == 1) return static_cast<float>(*reinterpret_cast<const int8_t*>(p));
      if (fd.size == 2) { int16_t v; std::memcpy(&v, p, 2); return static_cast<float>(v); }
      if (fd.size == 4) { int32_t v; std::memcpy(&v, p, 4); return static_cast<float>(v); }
      if (fd.size == 8) { int64_t v; std::memcpy(&v, p, 8); return static_cast<float>(v); }
      break;
  }
  return std::nanf("");
}

// LZF (Marc Lehmann's format, the one PCL embeds): literal runs and back references
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32) {  // literal run of ctrl + 1 bytes
      const size_t run = ctrl + 1;
      if (op + run > out_len || ip + run > in_len) return false;
      std::memcpy(out + op, in + ip, run);
      ip += run;
      op += run;
    } else {  // back reference
      size_t len = ctrl >> 5;
      if (ip >= in_len) return false;
      if (len == 7) {
        len += in[ip++];
        if (ip >= in_len) return false;
      }
      const size_t dist = ((ctrl & 0x1f) << 8) + in[ip++] + 1;
      len += 2;
      if (dist > op || op + len > out_len) return false;
      for (size_t k = 0; k < len; k++, op++) out[op] = out[op - dist];  // may overlap: byte by byte
    }
  }
  return op == out_len;
}

int find_field(const Header& h, const char* name) {
  for (size_t i = 0; i < h.fields.size(); i++)
    if (h.fields[i].name == name) return static_cast<int>(i);
  return -1;
}

}  // namespace

int pcd_read_header(const char* path, size_t* n_points, int* n_fields, int* data_kind, std::string& err) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path + ": " + std::strerror(errno); return 1; }
  Header h;
  err = parse_header(f, h);
  std::fclose(f);
  if (!err.empty()) { err = std::string(path) + ": " + err; return 2; }
  if (n_points) *n_points = h.points;
  if (n_fields) *n_fields = static_cast<int>(h.fields.size());
  if (data_kind) *data_kind = h.data;
  return 0;
}

int pcd_read_xyz(const char* path, void* out, size_t capacity, size_t stride, size_t* n_points, int* is_dense,
                 std::string& err) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path + ": " + std::strerror(errno); return 1; }
  Header h;
  err = parse_header(f, h);
  if (!err.empty()) { std::fclose(f); err = std::string(path) + ": " + err; return 2; }
  const int ix = find_field(h, "x"), iy = find_field(h, "y"), iz = find_field(h, "z");
  if (ix < 0 || iy < 0 || iz < 0) { std::fclose(f); err = std::string(path) + ": no x / y / z fields"; return 2; }
  if (h.fields[ix].count < 1 || h.fields[iy].count < 1 || h.fields[iz].count < 1) {
    std::fclose(f);
    err = std::string(path) + ": x / y / z with COUNT 0";
    return 2;
  }
  if (n_points) *n_points = h.points;
  if (h.points > capacity) { std::fclose(f); err = "output buffer too small"; return 3; }
  unsigned char* o = static_cast<unsigned char*>(out);
  bool dense = true;
  auto put = [&](size_t i, float x, float y, float z) {
    float rec[4] = {x, y, z, 1.0f};
    std::memcpy(o + i * stride, rec, stride >= 16 ? 16 : 12);
    if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) dense = false;
  };
  int rc = 0;
  if (h.data == 0) {  // ascii: one point per line, fields in order, COUNT values each
    std::string line;
    size_t i = 0;
    // column of each field's first value
    std::vector<size_t> col(h.fields.size());
    size_t c = 0;
    for (size_t k = 0; k < h.fields.size(); k++) { col[k] = c; c += h.fields[k].count; }
    while (i < h.points && read_line(f, line)) {
      std::vector<std::string> tok = split(line);
      if (tok.empty()) continue;
      if (tok.size() < c) { rc = 2; err = std::string(path) + ": short ascii record"; break; }
      auto val = [&](int k) {
        const std::string& t = tok[col[k]];
        if (upper(t) == "NAN") return std::nanf("");
        return static_cast<float>(std::strtod(t.c_str(), nullptr));
      };
      put(i++, val(ix), val(iy), val(iz));
    }
    if (!rc && i != h.points) { rc = 2; err = std::string(path) + ": fewer ascii records than POINTS"; }
  } else {
    std::vector<unsigned char> body;
    const size_t raw_bytes = h.points * h.record;
    if (h.data == 1) {
      // chunks of records through one small buffer (a 2M-point scan is a 24 MB body: a body-sized temporary costs as much
      // in page faults as the read itself); records whose x / y / z are 4-byte floats -- what every PCL writer produces
      // for PointXYZ* -- are copied without the per-scalar type dispatch
      const size_t chunk = 32768;
      // The body mapped where the page cache has it (NDT_PCD_MMAP=0: read through the chunk buffer): the read's copy out of
      // the page cache is half of the memory traffic of parsing a file, and the readers of a sequence share the host's
      // memory bandwidth (ndt_sequence.hpp).  A file that cannot be mapped (a pipe, an odd file system) is read.
      static const bool use_mmap = [] { const char* v = std::getenv("NDT_PCD_MMAP"); return !v || std::atoi(v) != 0; }();
      const unsigned char* mapped = nullptr;
      size_t map_len = 0, map_skew = 0;
      if (use_mmap && raw_bytes > 0 && h.body_bytes >= raw_bytes) {
        const long page = sysconf(_SC_PAGESIZE);
        map_skew = static_cast<size_t>(h.body_offset) % static_cast<size_t>(page);
        map_len = raw_bytes + map_skew;
        void* m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fileno(f), static_cast<off_t>(static_cast<size_t>(h.body_offset) - map_skew));
        if (m != MAP_FAILED) {
          mapped = static_cast<const unsigned char*>(m);
          (void)madvise(m, map_len, MADV_SEQUENTIAL);
        }
      }
      if (!mapped) body.resize(std::min(chunk, std::max<size_t>(h.points, 1)) * h.record);
      const Field &fx = h.fields[ix], &fy = h.fields[iy], &fz = h.fields[iz];
      const bool f32 = fx.type == 'F' && fx.size == 4 && fy.type == 'F' && fy.size == 4 && fz.type == 'F' && fz.size == 4;
      const bool wide = stride >= 16;
      uint32_t nonfinite = 0;
      for (size_t i0 = 0; i0 < h.points && !rc; i0 += chunk) {
        const size_t m = std::min(chunk, h.points - i0);
        if (!mapped && std::fread(body.data(), h.record, m, f) != m) { rc = 2; err = std::string(path) + ": truncated binary body"; break; }
        const unsigned char* r = mapped ? mapped + map_skew + i0 * h.record : body.data();
        unsigned char* d = o + i0 * stride;
        if (f32 && wide && h.record == 12 && fx.offset == 0 && fy.offset == 4 && fz.offset == 8) {
          // the plain "x y z" record of pcl::PCDWriter for PointXYZ: one unaligned 16-byte load per point (the last record of
          // the chunk is done by the general loop below: a 16-byte load there would read past the buffer)
          const __m128 one = _mm_castsi128_ps(_mm_set_epi32(0x3f800000, 0, 0, 0));
          const __m128 keep = _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1));
          const __m128i expo = _mm_set_epi32(0, 0x7f800000, 0x7f800000, 0x7f800000);
          __m128i bad = _mm_setzero_si128();
          size_t i = 0;
          for (; i + 1 < m; i++, r += 12, d += stride) {
            const __m128 v = _mm_or_ps(_mm_and_ps(_mm_loadu_ps(reinterpret_cast<const float*>(r)), keep), one);
            _mm_storeu_ps(reinterpret_cast<float*>(d), v);
            bad = _mm_or_si128(bad, _mm_cmpeq_epi32(_mm_and_si128(_mm_castps_si128(v), expo), expo));
          }
          if (_mm_movemask_epi8(bad) & 0x0fff) nonfinite = 1;
          for (; i < m; i++, r += 12, d += stride) {  // (the file's last record: 16 bytes from there could leave the mapping)
            uint32_t v[4];
            std::memcpy(v, r, 12);
            v[3] = 0x3f800000u;
            std::memcpy(d, v, 16);
            nonfinite |= static_cast<uint32_t>((v[0] & 0x7f800000u) == 0x7f800000u) | static_cast<uint32_t>((v[1] & 0x7f800000u) == 0x7f800000u) |
                         static_cast<uint32_t>((v[2] & 0x7f800000u) == 0x7f800000u);
          }
        } else if (f32) {
          for (size_t i = 0; i < m; i++, r += h.record, d += stride) {
            uint32_t v[4];
            std::memcpy(&v[0], r + fx.offset, 4);
            std::memcpy(&v[1], r + fy.offset, 4);
            std::memcpy(&v[2], r + fz.offset, 4);
            v[3] = 0x3f800000u;  // 1.0f
            std::memcpy(d, v, wide ? 16 : 12);
            // non-finite <=> exponent all ones
            nonfinite |= static_cast<uint32_t>((v[0] & 0x7f800000u) == 0x7f800000u) | static_cast<uint32_t>((v[1] & 0x7f800000u) == 0x7f800000u) |
                         static_cast<uint32_t>((v[2] & 0x7f800000u) == 0x7f800000u);
          }
        } else {
          for (size_t i = 0; i < m; i++, r += h.record)
            put(i0 + i, scalar_to_float(r + fx.offset, fx), scalar_to_float(r + fy.offset, fy), scalar_to_float(r + fz.offset, fz));
        }
      }
      if (nonfinite) dense = false;
      if (mapped) (void)munmap(const_cast<unsigned char*>(mapped), map_len);
    } else {
      uint32_t sizes[2];
      if (std::fread(sizes, 4, 2, f) != 2) { rc = 2; err = std::string(path) + ": truncated compressed header"; }
      if (!rc && sizes[1] != raw_bytes) { rc = 2; err = std::string(path) + ": uncompressed size does not match the header"; }
      if (!rc && sizes[0] > h.body_bytes) { rc = 2; err = std::string(path) + ": truncated compressed body"; }
      std::vector<unsigned char> comp;
      if (!rc) {
        comp.resize(sizes[0]);
        if (sizes[0] && std::fread(comp.data(), 1, sizes[0], f) != sizes[0]) { rc = 2; err = std::string(path) + ": truncated compressed body"; }
      }
      if (!rc) {
        body.resize(raw_bytes);
        if (!lzf_decompress(comp.data(), comp.size(), body.data(), raw_bytes)) { rc = 2; err = std::string(path) + ": corrupt LZF stream"; }
      }
      if (!rc) {  // struct of arrays: all of field 0, then all of field 1, ...
        std::vector<size_t> base(h.fields.size());
        size_t b = 0;
        for (size_t k = 0; k < h.fields.size(); k++) { base[k] = b; b += static_cast<size_t>(h.fields[k].size) * h.fields[k].count * h.points; }
        auto at = [&](int k, size_t i) {
          return scalar_to_float(body.data() + base[k] + i * static_cast<size_t>(h.fields[k].size) * h.fields[k].count, h.fields[k]);
        };
        for (size_t i = 0; i < h.points; i++) put(i, at(ix, i), at(iy, i), at(iz, i));
      }
    }
  }
  std::fclose(f);
  if (is_dense) *is_dense = dense ? 1 : 0;
  return rc;
}

int pcd_write_xyz(const char* path, const void* pts, size_t n, size_t stride, int binary, std::string& err) {
  FILE* f = std::fopen(path, "wb");
  if (!f) { err = std::string("cannot create ") + path + ": " + std::strerror(errno); return 1; }
  // header as pcl::PCDWriter::generateHeader writes it for PointXYZ
  std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                  "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA %s\n", n, n, binary ? "binary" : "ascii");
  const unsigned char* p = static_cast<const unsigned char*>(pts);
  bool ok = true;
  if (binary) {
    std::vector<float> buf(n * 3);
    for (size_t i = 0; i < n; i++) std::memcpy(&buf[i * 3], p + i * stride, 12);
    ok = n == 0 || std::fwrite(buf.data(), 12, n, f) == n;
  } else {
    for (size_t i = 0; i < n && ok; i++) {
      float v[3];
      std::memcpy(v, p + i * stride, 12);
      for (int k = 0; k < 3; k++) {
        if (std::isnan(v[k])) ok = std::fputs("nan", f) >= 0;
        else ok = std::fprintf(f, "%.8g", static_cast<double>(v[k])) > 0;  // PCDWriter's default precision
        std::fputc(k == 2 ? '\n' : ' ', f);
      }
    }
  }
  if (std::fclose(f) != 0) ok = false;
  if (!ok) { err = std::string("write failed: ") + path; return 2; }
  return 0;
}

}  // namespace ndt
