// ndt_latency.hip -- the single-scan latency path: one-launch evaluation (derivatives + final
// reduction + publication) and the persistent evaluation server.  Built with -fno-slp-vectorize:
// the SLP vectoriser packs the f32 neighbour math into v_pk_* pairs at the price of register
// shuffling, which costs these latency-bound kernels ~5 % (measured A/B on one MI355X) while the
// throughput kernels of ndt_kernels.hip gain from it.
#include "ndt_device.hpp"

#include <emmintrin.h>

namespace ndt {

namespace {

// ---------------------------------------------------------------------------
// Single-scan latency path: derivatives + final reduction + publication in ONE launch.
//
// Every block stores its 32-f64 partial row write-through (sc1), drains it (s_waitcnt vmcnt(0))
// and takes a ticket with one relaxed agent-scope fetch_add; the block whose ticket is the last
// re-reads ALL rows with sc1 loads (L1 is bypassed; every row was written through before its
// block's ticket), sums them in a fixed order and writes the packed row plus the sequence word
// straight into pinned host memory.  This is the ticket form of the hand-off of
// cdna_hip_programming.md Guideline 16 (sc1 stores / sc1 loads / drained before the counter add /
// last arriver told by the value its add returned; other waves of the last block load only after
// the workgroup barrier that the ticket wave joins).  Saves the second launch and the
// inter-kernel gap of the two-kernel path (~5 us per evaluation at 100k points).
// The counter is reset by the last block, so it is 0 again at the next launch.
// ---------------------------------------------------------------------------
#ifdef NDT_DIAG_BLOCK_CLOCKS  // (diagnostic build: when every block of the one-launch kernel started and its waves ended)
__device__ unsigned long long g_block_clocks[4096][4];
__device__ __forceinline__ unsigned long long wall_ticks() {  // the constant 100 MHz clock (s_memtime is per shader engine)
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#endif
constexpr int kFusedShards = 16;  // shard counters of the fused kernel's ticket: counter[32 * (1 + s)], top counter[0]
template <int NNB, bool WANT_H, int TPB>
__global__ __launch_bounds__(TPB) void k_derivatives_fused(const float4* __restrict__ src, int n, GridView gv, EvalParams P,
                                                           double* __restrict__ partials, unsigned* __restrict__ counter,
                                                           double* __restrict__ out_row, unsigned long long seq, int ppb) {
  constexpr int kWaves = TPB / kWave, kParts = TPB / kEvalStride;
  __shared__ double lds[kWaves * 32];
  __shared__ double lds2[kParts * kEvalStride];
  __shared__ int s_last;
  // The 81 parameter words go to LDS first: read from the kernel arguments they occupy ~90 SGPRs,
  // which spill to VGPR lanes (v_readlane / v_writelane made up ~20 % of this kernel's VALU count).
  __shared__ EvalParams sP;
  __shared__ PackedTables sT;
  {
    const int* sp = reinterpret_cast<const int*>(&P);
    int* dp = reinterpret_cast<int*>(&sP);
    for (int t = threadIdx.x; t < static_cast<int>(sizeof(EvalParams) / 4); t += TPB) dp[t] = sp[t];
    pack_tables(P, sT, threadIdx.x, TPB);
  }
  __syncthreads();
#ifdef NDT_DIAG_BLOCK_CLOCKS
  __shared__ unsigned long long s_wave_end[kWaves];
  const unsigned long long t_block_start = wall_ticks();
#endif
  double acc[kNumAcc];
#pragma unroll
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
  // XCD-aware first tile (see xcd_chunk); larger scans continue grid-strided, which balances uneven
  // neighbour counts better than one contiguous range per block (measured: -8 % at 2M points)
  // ppb points per block (points_per_block(n)): small scans use only the first ppb lanes of a block, see there
  const int first = (static_cast<int>(threadIdx.x) < ppb) ? xcd_chunk(blockIdx.x, gridDim.x) * ppb + static_cast<int>(threadIdx.x) : n;
  if (NNB == 27) derivatives_body_kd<WANT_H>(src, n, gv, sP, sT, first, gridDim.x * ppb, acc);
  else derivatives_body<NNB == 27 ? 7 : NNB, WANT_H, false, false, true, true>(src, n, gv, sP, sT, first, gridDim.x * ppb, acc);  // (LIMIT, REC4: 126 VGPRs)

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#ifdef NDT_DIAG_BLOCK_CLOCKS
  if (lane == 0) s_wave_end[wave] = wall_ticks();
#endif
  const double tot = wave_fold<kNumAcc>(acc);
  if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
  __syncthreads();
#ifdef NDT_DIAG_BLOCK_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < kWaves; w++) { lo = min(lo, s_wave_end[w]); hi = max(hi, s_wave_end[w]); }
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_block_clocks[blockIdx.x][0] = t_block_start;
    g_block_clocks[blockIdx.x][1] = lo;
    g_block_clocks[blockIdx.x][2] = hi;
    g_block_clocks[blockIdx.x][3] = xcc;
  }
#endif
  if (wave == 0) {
    if (lane < kEvalStride) {
      double v = 0.0;
      if (lane < kNumAcc) {
        v = lds[lane];
#pragma unroll
        for (int w = 1; w < kWaves; w++) v += lds[w * 32 + lane];
      }
      // write-through store of the whole 256-B row by one wave instruction
      __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * kEvalStride + lane, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      // Two-level ticket: shard s = the blocks b with b % kFusedShards == s, a counter per shard 128 B apart, and a top
      // counter for the shards' last arrivers.  (Returning atomics on ONE word serialise at the memory side at ~15 ns each:
      // the 512 blocks of a multi-million-point scan queued ~7 us behind a single counter, every evaluation.)  Every
      // counter is left at zero by the block that takes its last ticket.
      // (A scan that takes its blocks several passes -- multi-million points -- has them finish spread over tens of
      // microseconds: there the second level only adds its own round trip, measured -1 % at 2 M points, so one counter.)
      const bool one_pass = static_cast<long long>(gridDim.x) * ppb >= n;
      int last = 0;
      if (one_pass && gridDim.x > 1u) {
        const unsigned shards = min(static_cast<unsigned>(kFusedShards), gridDim.x);
        const unsigned shard = blockIdx.x % shards;
        const unsigned in_shard = (gridDim.x + shards - 1u - shard) / shards;
        unsigned* const c1 = counter + 32u * (1u + shard);
        if (__hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1u) {
          __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          last = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u) ? 1 : 0;
        }
      } else {
        last = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1 : 0;
      }
      s_last = last;
    }
  }
  __syncthreads();
  if (!s_last) return;

  // last arriver: fixed-order sum of all rows, every load sc1
  const int k = threadIdx.x % kEvalStride, part = threadIdx.x / kEvalStride;
  lds2[part * kEvalStride + k] = sum_rows_fixed<kParts>(partials, gridDim.x, threadIdx.x);
  __syncthreads();
  if (threadIdx.x < kEvalStride) {
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < kParts; p++) t += lds2[p * kEvalStride + threadIdx.x];
    lds[threadIdx.x] = t;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  publish_row_tagged(out_row, lds, threadIdx.x, seq);
}

// ---------------------------------------------------------------------------
// Persistent evaluation server (single-scan latency path).
//
// One launch per align(): gridDim.x resident blocks loop { wait for a command; evaluate; publish }.
// The host posts (sequence number, kind, EvalParams) into its mailbox.  Two placements of that mailbox:
//   direct  (default on large-BAR systems): the mailbox is fine-grained DEVICE memory the CPU writes through
//           the BAR; every block reads the command there itself (system-scope loads of its own HBM);
//   relayed (NDT_MAILBOX=host or NDT_SERVER_DIRECT=0): the mailbox is pinned HOST memory (or device memory),
//           wave 0 of block 0 polls it and relays the command into a second, device-side mailbox
//           (write-through) that all blocks poll with L1-bypassing loads -- one poller on the PCIe side.
// Measured: relayed/host 0.496 ms per headline registration, relayed/device 0.478 ms, direct 0.464 ms.
// Per evaluation this removes the kernel launch, the dispatch latency and the kernel-boundary cache
// invalidation (the read-only source / LUT / records stay L2-warm across evaluations).
//
// Liveness: every spin is bounded by a wall-clock budget (s_memrealtime, 100 MHz).  If no command
// arrives within `idle_ticks` the relay broadcasts EXIT and raises the host-visible `dead` word; a
// worker that sees no command for 4x that budget leaves on its own.  The grid therefore always
// drains, whatever the host does.  gridDim.x must not exceed the number of co-resident blocks.
// ---------------------------------------------------------------------------
constexpr int kServerTPB = 512;
static_assert(kServerTPB / kEvalStride == kServerParts, "the host adds kServerParts part sums");
constexpr int kCmdExit = 0x7fffffff;
constexpr int kCmdTransformExit = 4;  // transform the source by T into the output cloud, then exit

// Command = 32 self-validating 8-byte words in pinned host memory (same format in the device
// mailbox): word = (32 payload bits << 32) | (low 32 bits of the command's sequence number).
//   words  0..11  T[12]   (3x4 f32 transform)
//   word   12     kind    (0 with Hessian, 1 without, 2 f64 Hessian, 3 no-op, 4 transform + exit, EXIT)
//   words 13..24  cos/sin of roll, pitch, yaw after the 1e-4 snap: 6 f64 as (low, high) word pairs
//   words 25..31  zero
// A reader accepts the command when all 32 words carry the expected tag, so nothing depends on how
// the CPU's stores or the relay's 32-lane store are split into bus transactions (an aligned 8-byte
// word is single-copy atomic on both sides).  The host fills it with non-temporal stores (full-line
// writes, no read-for-ownership, so the CPU never fights the device's polling reads for the lines);
// the relay's poll (one 32-lane load) IS the data read, and it forwards the words with one store.
// The 69 angle-derivative coefficients (computeAngleDerivatives, ndt_omp_impl.hpp:288-395) are a
// function of the six cos/sin values; every block recomputes them (bit-identical to the host's: same
// f64 inputs, same operation order, contraction off) instead of fetching 344 B of tables.
// (Measured alternatives that were slower: parameter image + separate sequence word, two more
// dependent round trips, +4 us per command; a polled line written with ordinary stores, +8 us.)
constexpr int kCmdWords = 32;
struct ServerMailbox {
  unsigned long long cmd[kCmdWords];
  unsigned long long dead;  // host mailbox only: server gave up waiting (own line)
  unsigned long long pad[15];
};

// 69 entries (j_ang 8x3 then h_ang 15x3): value = s1*f[a1]*f[b1]*f[c1] + s2*f[a2]*f[b2]*f[c2],
// f = {1, sx, cx, sy, cy, sz, cz}; generated from the expressions of ndt_omp_impl.hpp:329-393
__device__ __constant__ signed char kAngleTerms[69][8] = {
    {-1, 1, 5, 0, 1, 2, 3, 6}, {-1, 1, 6, 0, -1, 2, 3, 5}, {-1, 2, 4, 0, 0, 0, 0, 0},
    {1, 2, 5, 0, 1, 1, 3, 6}, {1, 2, 6, 0, -1, 1, 3, 5}, {-1, 1, 4, 0, 0, 0, 0, 0},
    {-1, 3, 6, 0, 0, 0, 0, 0}, {1, 3, 5, 0, 0, 0, 0, 0}, {1, 4, 0, 0, 0, 0, 0, 0},
    {1, 1, 4, 6, 0, 0, 0, 0}, {-1, 1, 4, 5, 0, 0, 0, 0}, {1, 1, 3, 0, 0, 0, 0, 0},
    {-1, 2, 4, 6, 0, 0, 0, 0}, {1, 2, 4, 5, 0, 0, 0, 0}, {-1, 2, 3, 0, 0, 0, 0, 0},
    {-1, 4, 5, 0, 0, 0, 0, 0}, {-1, 4, 6, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {1, 2, 6, 0, -1, 1, 3, 5}, {-1, 2, 5, 0, -1, 1, 3, 6}, {0, 0, 0, 0, 0, 0, 0, 0},
    {1, 1, 6, 0, 1, 2, 3, 5}, {1, 2, 3, 6, -1, 1, 5, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {-1, 2, 5, 0, -1, 1, 3, 6}, {-1, 2, 6, 0, 1, 1, 3, 5}, {1, 1, 4, 0, 0, 0, 0, 0},
    {-1, 1, 5, 0, 1, 2, 3, 6}, {-1, 2, 3, 5, -1, 1, 6, 0}, {-1, 2, 4, 0, 0, 0, 0, 0},
    {1, 2, 4, 6, 0, 0, 0, 0}, {-1, 2, 4, 5, 0, 0, 0, 0}, {1, 2, 3, 0, 0, 0, 0, 0},
    {1, 1, 4, 6, 0, 0, 0, 0}, {-1, 1, 4, 5, 0, 0, 0, 0}, {1, 1, 3, 0, 0, 0, 0, 0},
    {-1, 1, 6, 0, -1, 2, 3, 5}, {1, 1, 5, 0, -1, 2, 3, 6}, {0, 0, 0, 0, 0, 0, 0, 0},
    {1, 2, 6, 0, -1, 1, 3, 5}, {-1, 1, 3, 6, -1, 2, 5, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {-1, 4, 6, 0, 0, 0, 0, 0}, {1, 4, 5, 0, 0, 0, 0, 0}, {-1, 3, 0, 0, 0, 0, 0, 0},
    {-1, 1, 3, 6, 0, 0, 0, 0}, {1, 1, 3, 5, 0, 0, 0, 0}, {1, 1, 4, 0, 0, 0, 0, 0},
    {1, 2, 3, 6, 0, 0, 0, 0}, {-1, 2, 3, 5, 0, 0, 0, 0}, {-1, 2, 4, 0, 0, 0, 0, 0},
    {1, 3, 5, 0, 0, 0, 0, 0}, {1, 3, 6, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {-1, 1, 4, 5, 0, 0, 0, 0}, {-1, 1, 4, 6, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {1, 2, 4, 5, 0, 0, 0, 0}, {1, 2, 4, 6, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {-1, 4, 6, 0, 0, 0, 0, 0}, {1, 4, 5, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {-1, 2, 5, 0, -1, 1, 3, 6}, {-1, 2, 6, 0, 1, 1, 3, 5}, {0, 0, 0, 0, 0, 0, 0, 0},
    {-1, 1, 5, 0, 1, 2, 3, 6}, {-1, 2, 3, 5, -1, 1, 6, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
};

// one coefficient of the f64 vectors j_ang_* (e < 24) / h_ang_* (e >= 24) from f = {1,sx,cx,sy,cy,sz,cz}
__device__ __forceinline__ double angle_coefficient_terms(const signed char* t, const double* f) {
#pragma clang fp contract(off)
  const double t1 = ((static_cast<double>(t[0]) * f[t[1]]) * f[t[2]]) * f[t[3]];
  const double t2 = ((static_cast<double>(t[4]) * f[t[5]]) * f[t[6]]) * f[t[7]];
  return t1 + t2;
}
__device__ __forceinline__ double angle_coefficient_f64(int e, const double* f) { return angle_coefficient_terms(kAngleTerms[e], f); }
// the same from a thread's own copy of its table row (the server keeps it in two registers across rounds: the row is
// fixed per thread, and its load from constant memory headed every round's table phase)
__device__ __forceinline__ double angle_coefficient_f64_held(unsigned lo, unsigned hi, const double* f) {
  signed char t[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t[k] = static_cast<signed char>((lo >> (8 * k)) & 0xff);
    t[4 + k] = static_cast<signed char>((hi >> (8 * k)) & 0xff);
  }
  return angle_coefficient_terms(t, f);
}
// the f32 matrices j_ang / h_ang hold the same values rounded, except h_ang row d1, z: +sy (:383)
// where the f64 vector has -sy (:361)
__device__ __forceinline__ float angle_coefficient(int e, const double* f) {
  const double v = (e == 24 + 6 * 3 + 2) ? f[3] : angle_coefficient_f64(e, f);
  return static_cast<float>(v);
}

template <int NNB>
__global__ __launch_bounds__(kServerTPB) void k_eval_server(const float4* __restrict__ src, int n, GridView gv,
                                                            ServerMailbox* host_mb, ServerMailbox* dev_mb,
                                                            double* __restrict__ partials, unsigned* __restrict__ counter,
                                                            double* __restrict__ out_row, unsigned long long first_seq,
                                                            unsigned long long idle_ticks, double gauss_d1, double gauss_d2,
                                                            int param_pad, const float4* __restrict__ out_src,
                                                            float4* __restrict__ out_dst, int out_n, unsigned long long* dbg,
                                                            int direct, float4* __restrict__ out_host, int ppb,
                                                            unsigned* __restrict__ counter_next) {
  constexpr int kWaves = kServerTPB / kWave, kParts = kServerTPB / kEvalStride;
  // the shard counters of the NEXT launch (the other of two sets: nobody touches it during this one) start at zero --
  // instead of a fill kernel in front of every registration's server
  if (blockIdx.x == 0 && threadIdx.x < 1 + kParts) counter_next[32u * threadIdx.x] = 0u;
  __shared__ double lds[kWaves * 32];
  __shared__ EvalParams sP;
  __shared__ Hess64Params sP64;
  __shared__ PackedTables sT;
  __shared__ double s_f[8];  // 1, sx, cx, sy, cy, sz, cz
  __shared__ int s_kind;
  unsigned long long expect = first_seq;
  // this lane's first point of every round (see derivatives_body PRELOADED)
  const int my_first = (static_cast<int>(threadIdx.x) < ppb) ? xcd_chunk(blockIdx.x, gridDim.x) * ppb + static_cast<int>(threadIdx.x) : n;
  float4 my_pt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (my_first < n) my_pt = src[my_first];
  unsigned terms_lo = 0, terms_hi = 0;  // this thread's row of kAngleTerms (threads 0..68 build the angle tables)
  int pack_pos = 0;                     // ... and where its coefficient goes in PackedTables
  if (threadIdx.x < 69) {
    pack_pos = kPackPos[threadIdx.x];
    const signed char* t = kAngleTerms[threadIdx.x];
    for (int k = 0; k < 4; k++) {
      terms_lo |= static_cast<unsigned>(static_cast<unsigned char>(t[k])) << (8 * k);
      terms_hi |= static_cast<unsigned>(static_cast<unsigned char>(t[4 + k])) << (8 * k);
    }
  }
  if (threadIdx.x == 0) {
    sP.d1 = gauss_d1;
    sP.d2 = static_cast<float>(gauss_d2);
    sP.pad = param_pad;
    sP64.d1 = gauss_d1;
    sP64.d2 = gauss_d2;
    sP64.r2 = static_cast<double>(__int_as_float(param_pad));
  }

  for (;;) {
    // Nothing but `expect` is meant to live across rounds: opaque copies keep the compiler from
    // hoisting per-round address arithmetic out of the loop (it did, ran out of registers and
    // spilled those values to scratch, whose reloads sat on the round's critical path).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    asm volatile("" : "+s"(host_mb), "+s"(dev_mb), "+s"(partials), "+s"(counter), "+s"(out_row), "+s"(dbg), "+s"(src));
    // ---- relay: host mailbox -> device mailbox (wave 0 of block 0) ----
    if (!direct && blockIdx.x == 0 && wave == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long w = 0;
      bool got = false;
      for (;;) {  // the poll IS the data read
        if (lane < kCmdWords) w = __hip_atomic_load(&host_mb->cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__ballot(lane >= kCmdWords || static_cast<unsigned>(w) == static_cast<unsigned>(expect)) == ~0ull) { got = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) break;
      }
      const unsigned long long dbg_seen = __builtin_amdgcn_s_memrealtime();
      if (!got) {  // idle for too long: tell the host, send everybody home
        if (lane == 0) __hip_atomic_store(&host_mb->dead, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w = tag_word(lane == 12 ? static_cast<unsigned>(kCmdExit) : 0u, expect);
      }
      if (lane < kCmdWords) __hip_atomic_store(&dev_mb->cmd[lane], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dbg && got && static_cast<int>(__shfl(static_cast<unsigned>(w >> 32), 12, kWave)) != kCmdExit && lane == 0) { dbg[0] = dbg_seen; dbg[1] = __builtin_amdgcn_s_memrealtime(); }  // seen / relayed
    }
    // ---- every block: wait for the device command block ----
    if (wave == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long w = 0;
      bool got = false;
      if (direct) {
        // the host's mailbox is in this device's own (host-visible) memory: every block reads the command where
        // the CPU put it -- no relay hop.  Block 0 doubles as the watchdog that tells the host when the server
        // gives up; the others allow four times its patience, as in the relayed mode.
        const unsigned long long patience = (blockIdx.x == 0) ? idle_ticks : 4 * idle_ticks;
        for (;;) {
          if (lane < kCmdWords) w = __hip_atomic_load(&host_mb->cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else if (lane == kCmdWords) w = __hip_atomic_load(&host_mb->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (__ballot(lane >= kCmdWords || static_cast<unsigned>(w) == static_cast<unsigned>(expect)) == ~0ull) { got = true; break; }
          if (__ballot(lane == kCmdWords && w == expect) != 0) break;  // block 0 gave up on this very command: leave with it
          if (__builtin_amdgcn_s_memrealtime() - t0 > patience) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (!got && blockIdx.x == 0 && lane == 0) __hip_atomic_store(&host_mb->dead, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else
      for (;;) {
        if (lane < kCmdWords) w = __hip_atomic_load(&dev_mb->cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(lane >= kCmdWords || static_cast<unsigned>(w) == static_cast<unsigned>(expect)) == ~0ull) { got = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 4 * idle_ticks) break;
        __builtin_amdgcn_s_sleep(1);
      }
      int kind = kCmdExit;
      if (got) {
        const unsigned payload = static_cast<unsigned>(w >> 32);
        const unsigned next = __shfl_down(payload, 1, kWave);
        kind = static_cast<int>(__shfl(payload, 12, kWave));
        if (lane < 12) {  // T[12]
          const float t = __int_as_float(static_cast<int>(payload));
          sP.T[lane] = t;
          sP64.T[lane] = t;
        }
        if (lane >= 13 && lane < 25 && ((lane - 13) & 1) == 0) {  // cx cy cz sx sy sz -> f = {1, sx, cx, sy, cy, sz, cz}
          const double v = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(next) << 32) | payload));
          const int a = (lane - 13) >> 1;  // 0..2 cos, 3..5 sin
          s_f[(a < 3) ? 2 + 2 * a : 1 + 2 * (a - 3)] = v;
        }
        if (lane == 0) { s_f[0] = 1.0; s_f[7] = 0.0; }
      }
      if (lane == 0) {
        s_kind = kind;
        if (dbg && kind != kCmdExit) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          dbg[8 + 2 * blockIdx.x] = now;  // block has its command
          if (direct && blockIdx.x == 0 && kind != kCmdTransformExit) dbg[0] = dbg[1] = now;  // direct mailbox: time zero of the diagnostics
        }
      }
    }
    __syncthreads();
    const int kind = s_kind;
    if (kind == kCmdTransformExit) {  // last command of a registration: write the aligned cloud, then leave
      for (int i = blockIdx.x * kServerTPB + tid; i < out_n; i += gridDim.x * kServerTPB) {
        const float4 pt = out_src[i];
        float tx, ty, tz;
        xform_point(sP.T, pt.x, pt.y, pt.z, tx, ty, tz);
        const float4 moved = make_float4(tx, ty, tz, 1.0f);
        out_dst[i] = moved;
        if (out_host) out_host[i] = moved;  // the caller wants the cloud on the host: written there directly (page-locked)
      }
      return;
    }
    if (kind < 0 || kind > 3) return;  // EXIT or time-out: the whole block leaves together (3 = no-op round)
    if (tid < 69) {
      const double c64 = angle_coefficient_f64_held(terms_lo, terms_hi, s_f);
      if (kind == 2) {  // f64 vectors of computeHessian (:329-361, -sy in row d1)
        if (tid < 24) sP64.jd[tid / 3][tid % 3] = c64;
        else sP64.hd[(tid - 24) / 3][(tid - 24) % 3] = c64;
      } else {  // the f32 matrices hold the same values rounded, except h_ang row d1, z: +sy (:383)
        const float c = static_cast<float>((tid == 24 + 6 * 3 + 2) ? s_f[3] : c64);
        reinterpret_cast<float*>(&sT)[pack_pos] = c;  // straight into the pair layout of the packed math
      }
    }
    __syncthreads();
    unsigned long long* fine = dbg ? dbg + 8 + 2 * 1024 + 8 * blockIdx.x : nullptr;  // diagnostics: per-block phase stamps
    if (fine && tid == 0) fine[0] = __builtin_amdgcn_s_memrealtime();

    // ---- evaluate ----
    double acc[kNumAcc];
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
    // XCD-aware first tile (see xcd_chunk); larger scans continue grid-strided (better balance)
    const int first = (tid < ppb) ? xcd_chunk(blockIdx.x, gridDim.x) * ppb + tid : n, stride = gridDim.x * ppb;
    const int limit = n;
    if (kind == 2) {
      // rare round (at most one per Newton iteration): keep its loop invariants from being hoisted
      // into registers the hot rounds need (the opaque copy of `first` pins them inside the branch)
      int first64 = first;
      asm volatile("" : "+v"(first64));
      hessian64_body<NNB, true>(src, limit, gv, sP64, first64, stride, acc);
    } else if (NNB == 27) {
      if (kind == 0) derivatives_body_kd<true>(src, limit, gv, sP, sT, first, stride, acc);
      else if (kind == 1) derivatives_body_kd<false>(src, limit, gv, sP, sT, first, stride, acc);
    } else {
      if (kind == 0) derivatives_body<NNB == 27 ? 7 : NNB, true, false, true>(src, limit, gv, sP, sT, first, stride, acc, nullptr, my_pt);
      else if (kind == 1) derivatives_body<NNB == 27 ? 7 : NNB, false, false, true>(src, limit, gv, sP, sT, first, stride, acc, nullptr, my_pt);
    }
    if (fine && tid == 0) fine[1] = __builtin_amdgcn_s_memrealtime();
    const double tot = wave_fold<kNumAcc>(acc);
    if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
    if (fine && tid == 0) fine[2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (wave == 0) {
      if (fine && lane == 0) fine[3] = __builtin_amdgcn_s_memrealtime();
      if (lane < kEvalStride) {
        double v = 0.0;
        if (lane < kNumAcc) {
          v = lds[lane];
#pragma unroll
          for (int w = 1; w < kWaves; w++) v += lds[w * 32 + lane];
        }
        __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * kEvalStride + lane, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (fine && lane == 0) fine[4] = __builtin_amdgcn_s_memrealtime();
      // Fan-in over kParts = 16 shard counters: shard p = the blocks b with b % 16 == p, exactly the rows that part p
      // of the fixed-order sum (sum_rows_fixed) adds.  The last arriver of a shard adds its shard's rows and publishes
      // that PART sum to the host, which adds the 16 parts in order -- the same additions in the same order as one
      // last block doing both stages, without the top-level ticket and the second stage on the device
      // (~200 returning atomics on ONE word would serialise at ~13 ns each; a shard sees 12-13).
      // Counters live 128 B apart and are never reset inside a launch.  All of it stays inside this wave: the ticket's
      // answer, the part sum and its publication (lane to lane by shuffles) need neither LDS nor the block's other
      // waves, which wait at the round's last barrier (one block barrier and an LDS round trip less per evaluation).
      const unsigned round = static_cast<unsigned>(expect - first_seq);
      const unsigned shard = blockIdx.x % static_cast<unsigned>(kParts);
      const unsigned in_shard = (gridDim.x + static_cast<unsigned>(kParts) - 1u - shard) / static_cast<unsigned>(kParts);
      unsigned t1 = 0u;
      if (lane == 0) {
        t1 = __hip_atomic_fetch_add(counter + 32u * (1u + shard), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dbg) dbg[9 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();  // block has its ticket
      }
      if (__builtin_amdgcn_readfirstlane(t1) == (round + 1u) * in_shard - 1u) {
        if (dbg && lane == 0)  // a shard's last arriver starts its part sum (the latest writer is the one that matters)
          __hip_atomic_store(&dbg[2], static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double part = (lane < kEvalStride) ? sum_rows_fixed<kParts>(partials, gridDim.x, static_cast<int>(shard) * kEvalStride + lane) : 0.0;
        publish_lanes_tagged(out_row + static_cast<size_t>(shard) * kPublishSlots, part, lane, expect);
        if (dbg && lane == 0)  // published
          __hip_atomic_store(&dbg[3], static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();  // s_kind / lds are rewritten by the next round
    expect++;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Diagnostic (never used by the product path): the evaluation server's round with the HOST taken out of the loop.
// Same tables, body, fold, row store, shard tickets and part sums as k_eval_server; then the 16 shard-last blocks
// meet on a second-level ticket, the last of them adds the 16 parts (what the host does today) and posts the NEXT
// command itself -- same pose, next sequence number -- into a device mailbox that every block polls.  No Newton /
// More-Thuente step is computed: kernel time / rounds is the protocol + body cost a device-side solver would START from
// (what moving the solver onto the device could save at best).
// ---------------------------------------------------------------------------------------------------------
template <int NNB>
__global__ __launch_bounds__(kServerTPB) void k_selfdrive(const float4* __restrict__ src, int n, GridView gv, ServerMailbox* dev_mb,
                                                          double* __restrict__ partials, unsigned* __restrict__ counter,
                                                          double* __restrict__ parts, double* __restrict__ out_row,
                                                          unsigned long long first_seq, int rounds, int with_body, double gauss_d1,
                                                          double gauss_d2, int param_pad, int ppb) {
  constexpr int kWaves = kServerTPB / kWave, kParts = kServerTPB / kEvalStride;
  __shared__ double lds[kWaves * 32];
  __shared__ EvalParams sP;
  __shared__ PackedTables sT;
  __shared__ double s_f[8];
  __shared__ int s_last;
  __shared__ int s_final;
  __shared__ unsigned long long s_cmd[kCmdWords];
  unsigned long long expect = first_seq;
  const int my_first = (static_cast<int>(threadIdx.x) < ppb) ? xcd_chunk(blockIdx.x, gridDim.x) * ppb + static_cast<int>(threadIdx.x) : n;
  float4 my_pt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (my_first < n) my_pt = src[my_first];
  unsigned terms_lo = 0, terms_hi = 0;
  int pack_pos = 0;
  if (threadIdx.x < 69) {
    pack_pos = kPackPos[threadIdx.x];
    const signed char* t = kAngleTerms[threadIdx.x];
    for (int k = 0; k < 4; k++) {
      terms_lo |= static_cast<unsigned>(static_cast<unsigned char>(t[k])) << (8 * k);
      terms_hi |= static_cast<unsigned>(static_cast<unsigned char>(t[4 + k])) << (8 * k);
    }
  }
  if (threadIdx.x == 0) {
    sP.d1 = gauss_d1;
    sP.d2 = static_cast<float>(gauss_d2);
    sP.pad = param_pad;
  }
  const unsigned n_parts = min(static_cast<unsigned>(kParts), gridDim.x);
  for (int round = 0; round < rounds; round++) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    asm volatile("" : "+s"(dev_mb), "+s"(partials), "+s"(counter), "+s"(parts), "+s"(out_row), "+s"(src));
    if (wave == 0) {  // every block: wait for the device command block
      unsigned long long w = 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        if (lane < kCmdWords) w = __hip_atomic_load(&dev_mb->cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(lane >= kCmdWords || static_cast<unsigned>(w) == static_cast<unsigned>(expect)) == ~0ull) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) break;  // 20 ms: never hang the grid
        __builtin_amdgcn_s_sleep(1);
      }
      const unsigned payload = static_cast<unsigned>(w >> 32);
      const unsigned next = __shfl_down(payload, 1, kWave);
      if (lane < kCmdWords) s_cmd[lane] = w;
      if (lane < 12) sP.T[lane] = __int_as_float(static_cast<int>(payload));
      if (lane >= 13 && lane < 25 && ((lane - 13) & 1) == 0) {
        const double v = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(next) << 32) | payload));
        const int a = (lane - 13) >> 1;
        s_f[(a < 3) ? 2 + 2 * a : 1 + 2 * (a - 3)] = v;
      }
      if (lane == 0) { s_f[0] = 1.0; s_f[7] = 0.0; }
    }
    __syncthreads();
    if (tid < 69) {
      const double c64 = angle_coefficient_f64_held(terms_lo, terms_hi, s_f);
      const float c = static_cast<float>((tid == 24 + 6 * 3 + 2) ? s_f[3] : c64);
      reinterpret_cast<float*>(&sT)[pack_pos] = c;
    }
    __syncthreads();
    double acc[kNumAcc];
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
    const int first = (tid < ppb) ? xcd_chunk(blockIdx.x, gridDim.x) * ppb + tid : n, stride = gridDim.x * ppb;
    if (with_body) derivatives_body<NNB, true, false, true>(src, n, gv, sP, sT, first, stride, acc, nullptr, my_pt);
    const double tot = wave_fold<kNumAcc>(acc);
    if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
    __syncthreads();
    if (wave == 0) {
      if (lane < kEvalStride) {
        double v = 0.0;
        if (lane < kNumAcc) {
          v = lds[lane];
#pragma unroll
          for (int w = 1; w < kWaves; w++) v += lds[w * 32 + lane];
        }
        __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * kEvalStride + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned shard = blockIdx.x % static_cast<unsigned>(kParts);
        const unsigned in_shard = (gridDim.x + static_cast<unsigned>(kParts) - 1u - shard) / static_cast<unsigned>(kParts);
        const unsigned t1 = __hip_atomic_fetch_add(counter + 32u * (1u + shard), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t1 == (static_cast<unsigned>(round) + 1u) * in_shard - 1u) ? 1 : 0;
        s_final = 0;
      }
    }
    __syncthreads();
    if (s_last) {  // part sum of this shard -> device memory, then the second-level ticket
      const int shard = static_cast<int>(blockIdx.x % static_cast<unsigned>(kParts));
      if (tid < kEvalStride) {
        const double v = sum_rows_fixed<kParts>(partials, gridDim.x, shard * kEvalStride + tid);
        __hip_atomic_store(parts + static_cast<size_t>(shard) * kEvalStride + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned t2 = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_final = (t2 == (static_cast<unsigned>(round) + 1u) * n_parts - 1u) ? 1 : 0;
      }
      __syncthreads();
      if (s_final) {  // what the host does today: the 16 parts in order ... and (here) the next command
        if (tid < kEvalStride) {
          double t = 0.0;
          for (unsigned p = 0; p < static_cast<unsigned>(kParts); p++)
            t += (p < n_parts) ? __hip_atomic_load(parts + static_cast<size_t>(p) * kEvalStride + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
          lds[tid] = t;
        }
        __syncthreads();
        if (round + 1 < rounds) {
          if (tid < kCmdWords)
            __hip_atomic_store(&dev_mb->cmd[tid], (s_cmd[tid] & 0xffffffff00000000ull) | ((expect + 1) & 0xffffffffull), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        } else {
          publish_row_tagged(out_row, lds, tid, expect);
        }
      }
    }
    __syncthreads();
    expect++;
  }
}

static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
}
}  // namespace

constexpr int kFusedTPB = 512;
// Points a block of the latency kernels works on.  A 512-thread block is eight waves, two per SIMD, and the body is
// VALU-issue-bound: two waves on a SIMD take ~4.4 us where one takes ~2.9.  A scan small enough to leave most of the
// chip empty anyway (the mapping nodes' real size: 16 k points = 32 full blocks on 256 CUs) therefore fills only the
// first half of each block's lanes -- twice the blocks, every busy wave with a SIMD of its own.
int points_per_block(int n, int cus) {
  static const int forced = env_int("NDT_K2_PPB", 0);
  if (forced >= 64 && forced <= 512 && forced % 8 == 0) return forced;
  // A block per CU while 512 points per block allow that (then 512, grid-strided): an evaluation's probes and record gathers
  // go through each CU's 64 B / clock vector memory path, and the fewer of them a CU has, the sooner its waves compute; with
  // up to 256 points per block every busy wave also has a SIMD of its own.  Measured (us per evaluation, set U): 16 k points
  // 10.7 at 256 points per block, 9.7 at 64; 30 k 11.4 -> 10.7 at 128; 48 k 12.0 -> 11.4 at 192; 100 k 12.2 at 512 (196
  // blocks) -> 11.6 at 392 (256 blocks).  More blocks than CUs is the one thing to avoid: 48 k at 128 per block, 14.3.
  if (cus <= 0) cus = 256;
  const int per_cu = ((n + cus - 1) / cus + 7) / 8 * 8;
  return std::max(64, std::min(512, per_cu));
}
int fused_blocks(int n, int cus, bool partition) {
  // two 512-thread blocks per CU (126 VGPRs) = every block resident at once on 256 CUs; measured on a 2M-point scan:
  // 256 / 384 / 512 / 640 / 768 / 1024 / 2048 blocks -> 449 / 446 / 538 / 465 / 464 / 488 / 432 registrations/s
  static const int cap = env_int("NDT_K2_MAX_BLOCKS", 512);
  const int ppb = points_per_block(n, cus);
  size_t b = (static_cast<size_t>(n) + ppb - 1) / ppb;
  if (b < 1) b = 1;
  if (b > static_cast<size_t>(cap)) b = cap;
  if (partition && cus > 0 && b > static_cast<size_t>(2 * cus)) b = 2 * cus;  // a handle on a CU partition: still all blocks resident at once
  return static_cast<int>(b);
}

hipError_t launch_derivatives_fused(const float4* src, int n, const GridView& gv, const EvalParams& P, int search,
                                    bool want_hessian, int n_blocks, int ppb, double* partials, unsigned* counter, double* out_row,
                                    unsigned long long seq, hipStream_t stream) {
#define NDT_LAUNCH_FUSED(NNB, H)                                                                                        \
  hipLaunchKernelGGL((k_derivatives_fused<NNB, H, kFusedTPB>), dim3(n_blocks), dim3(kFusedTPB), 0, stream, src, n, gv, P, \
                     partials, counter, out_row, seq, ppb)
  if (search == 0) {
    if (want_hessian) NDT_LAUNCH_FUSED(27, true); else NDT_LAUNCH_FUSED(27, false);
  } else if (search == 1) {
    if (want_hessian) NDT_LAUNCH_FUSED(26, true); else NDT_LAUNCH_FUSED(26, false);
  } else if (search == 3) {
    if (want_hessian) NDT_LAUNCH_FUSED(1, true); else NDT_LAUNCH_FUSED(1, false);
  } else {
    if (want_hessian) NDT_LAUNCH_FUSED(7, true); else NDT_LAUNCH_FUSED(7, false);
  }
#undef NDT_LAUNCH_FUSED
  return hipGetLastError();
}

#ifdef NDT_DIAG_BLOCK_CLOCKS
extern "C" int ndt_diag_block_clocks_read(unsigned long long* out, int n_blocks) {
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_clocks), static_cast<size_t>(n_blocks) * 4 * sizeof(unsigned long long)));
}
#endif
size_t server_mailbox_bytes() { return sizeof(ServerMailbox); }

// host side of the mailbox protocol (pinned, coherent host memory): the 32 tagged words are built
// locally and written with non-temporal 16-byte stores, full 64-byte lines
void server_post(void* host_mailbox, unsigned long long seq, int kind, const float* T12, const double* cos_sin6) {
  ServerMailbox* mb = static_cast<ServerMailbox*>(host_mailbox);
  alignas(64) unsigned long long c[kCmdWords];
  const unsigned long long tag = seq & 0xffffffffull;
  for (int i = 0; i < kCmdWords; i++) c[i] = tag;
  if (T12)
    for (int i = 0; i < 12; i++) {
      unsigned bits;
      std::memcpy(&bits, &T12[i], sizeof(bits));
      c[i] |= static_cast<unsigned long long>(bits) << 32;
    }
  c[12] |= static_cast<unsigned long long>(static_cast<unsigned>(kind)) << 32;
  if (cos_sin6)
    for (int i = 0; i < 6; i++) {
      unsigned long long bits;
      std::memcpy(&bits, &cos_sin6[i], sizeof(bits));
      c[13 + 2 * i] |= (bits & 0xffffffffull) << 32;
      c[14 + 2 * i] |= (bits >> 32) << 32;
    }
  for (int i = 0; i < kCmdWords / 2; i++)
    _mm_stream_si128(reinterpret_cast<__m128i*>(&mb->cmd[2 * i]), _mm_load_si128(reinterpret_cast<const __m128i*>(&c[2 * i])));
  _mm_sfence();
}
hipError_t launch_selfdrive(const float4* src, int n, const GridView& gv, int search, void* dev_mailbox, int n_blocks, int ppb, double* partials,
                            unsigned* counter, double* parts, double* out_row, unsigned long long first_seq, int rounds, int with_body,
                            double gauss_d1, double gauss_d2, int param_pad, hipStream_t stream) {
  if (search != 2) return hipErrorInvalidValue;  // DIRECT7 only: a diagnostic
  hipLaunchKernelGGL(k_selfdrive<7>, dim3(n_blocks), dim3(kServerTPB), 0, stream, src, n, gv, static_cast<ServerMailbox*>(dev_mailbox), partials,
                     counter, parts, out_row, first_seq, rounds, with_body, gauss_d1, gauss_d2, param_pad, ppb);
  return hipGetLastError();
}
unsigned long long server_dead_word(const void* host_mailbox) {
  return __atomic_load_n(&static_cast<const ServerMailbox*>(host_mailbox)->dead, __ATOMIC_ACQUIRE);
}
void server_reset_mailbox(void* host_mailbox) { std::memset(host_mailbox, 0, sizeof(ServerMailbox)); }

hipError_t launch_eval_server(const float4* src, int n, const GridView& gv, int search, void* host_mailbox,
                              void* dev_mailbox, int n_blocks, int ppb, double* partials, unsigned* counter, double* out_row,
                              unsigned long long first_seq, unsigned long long idle_ticks, double gauss_d1, double gauss_d2,
                              int param_pad, const float4* out_src, float4* out_dst, int out_n, hipStream_t stream,
                              unsigned long long* dbg, int direct, float4* out_host, unsigned* counter_next) {
  ServerMailbox* hm = static_cast<ServerMailbox*>(host_mailbox);
  ServerMailbox* dm = static_cast<ServerMailbox*>(dev_mailbox);
  if (search == 0)
    hipLaunchKernelGGL(k_eval_server<27>, dim3(n_blocks), dim3(kServerTPB), 0, stream, src, n, gv, hm, dm, partials, counter,
                       out_row, first_seq, idle_ticks, gauss_d1, gauss_d2, param_pad, out_src, out_dst, out_n, dbg, direct, out_host, ppb, counter_next);
  else if (search == 1)
    hipLaunchKernelGGL(k_eval_server<26>, dim3(n_blocks), dim3(kServerTPB), 0, stream, src, n, gv, hm, dm, partials, counter,
                       out_row, first_seq, idle_ticks, gauss_d1, gauss_d2, param_pad, out_src, out_dst, out_n, dbg, direct, out_host, ppb, counter_next);
  else if (search == 3)
    hipLaunchKernelGGL(k_eval_server<1>, dim3(n_blocks), dim3(kServerTPB), 0, stream, src, n, gv, hm, dm, partials, counter,
                       out_row, first_seq, idle_ticks, gauss_d1, gauss_d2, param_pad, out_src, out_dst, out_n, dbg, direct, out_host, ppb, counter_next);
  else
    hipLaunchKernelGGL(k_eval_server<7>, dim3(n_blocks), dim3(kServerTPB), 0, stream, src, n, gv, hm, dm, partials, counter,
                       out_row, first_seq, idle_ticks, gauss_d1, gauss_d2, param_pad, out_src, out_dst, out_n, dbg, direct, out_host, ppb, counter_next);
  return hipGetLastError();
}

}  // namespace ndt
