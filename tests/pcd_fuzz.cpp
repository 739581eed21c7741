// tests/pcd_fuzz.cpp -- mutation fuzzer of the PCD reader (ndt_pcd.cpp), built with AddressSanitizer +
// UndefinedBehaviorSanitizer by tests/test_pcd.py: truncated, spliced and bit-flipped files must be
// parsed or rejected, never crash.   pcd_fuzz [iterations] [scratch dir]
#include "ndt_pcd.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
int main(int argc, char** argv) {
  std::mt19937 rng(7);
  // base files: binary, ascii, compressed-literal
  std::vector<std::string> bases;
  {
    std::string h = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 50\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 50\nDATA binary\n";
    std::string b(50 * 16, '\0');
    for (auto& c : b) c = static_cast<char>(rng());
    bases.push_back(h + b);
    std::string a = "VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 5\nHEIGHT 1\nPOINTS 5\nDATA ascii\n1 2 3\n4 5 6\nnan 1 2\n7 8 9\n1e3 -2 0.5\n";
    bases.push_back(a);
    std::string ch = "VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 8\nHEIGHT 1\nPOINTS 8\nDATA binary_compressed\n";
    std::string raw(96, 'a');
    std::string comp;
    for (size_t i = 0; i < raw.size(); i += 32) { comp.push_back(31); comp.append(raw, i, 32); }
    unsigned sz[2] = {static_cast<unsigned>(comp.size()), 96};
    bases.push_back(ch + std::string(reinterpret_cast<char*>(sz), 8) + comp);
  }
  const std::string path_s = std::string(argc > 2 ? argv[2] : "/tmp") + "/fuzz_case.pcd";
  const char* path = path_s.c_str();
  int ok = 0, bad = 0;
  const int iters = argc > 1 ? std::atoi(argv[1]) : 5000;
  for (int it = 0; it < iters; it++) {
    std::string f = bases[it % bases.size()];
    const int nm = 1 + rng() % 6;
    for (int k = 0; k < nm; k++) {
      const int op = rng() % 4;
      if (f.empty()) break;
      const size_t pos = rng() % f.size();
      if (op == 0) f[pos] = static_cast<char>(rng());
      else if (op == 1) f.erase(pos, 1 + rng() % 40);
      else if (op == 2) f.insert(pos, std::string(1 + rng() % 8, static_cast<char>('0' + rng() % 10)));
      else f.resize(pos);
    }
    FILE* fp = std::fopen(path, "wb");
    std::fwrite(f.data(), 1, f.size(), fp);
    std::fclose(fp);
    size_t n = 0; int nf = 0, kind = 0, dense = 0; std::string err;
    if (ndt::pcd_read_header(path, &n, &nf, &kind, err) != 0) { bad++; continue; }
    if (n > 1000000) { bad++; continue; }  // the caller sizes the buffer from the header
    std::vector<float> out((n ? n : 1) * 4);
    if (ndt::pcd_read_xyz(path, out.data(), n, 16, &n, &dense, err) == 0) ok++; else bad++;
  }
  std::printf("fuzz: %d parsed, %d rejected, no crash\n", ok, bad);
  return 0;
}
