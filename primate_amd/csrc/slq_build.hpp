// slq_build.hpp — an operator's derived device data built ON the device (r04): the permuted CSR, the exact-symmetry
// check and the upper triangle, the per-tile line lists and the descriptor / record streams of the ring-fed passes
// (slq_ring.hpp). The host keeps what is sequential and small - the order (Cuthill-McKee in the chunks), the clusters and the
// runs of the upper-triangle tiles - and uploads the caller's CSR once, unpermuted, while it works on them.
//
// Every kernel here restates a host builder of slq.hip (named at each one) and produces the same bytes: SLQ_DEVICE_BUILD=2
// builds both ways and compares (tests/test_gpu_parity.py::test_device_built_streams_equal_the_host_built_ones). None of this is
// arithmetic of the Lanczos path: integer bookkeeping, one thread per row or per tile, a few hundred microseconds per operator.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "slq_common.hpp"

namespace slqb {
using namespace slq;

constexpr int kScanTile = 4096;  // elements of a scan block: 256 threads x 16

// ---- inclusive scan of a[0, count) in place (int32; the callers keep a[-1] = 0 in front: row pointers, tile offsets) ----
// three launches: sums of blocks of kScanTile, their scan by one workgroup, the blocks again with their offsets
__global__ __launch_bounds__(256) void k_scan_block_sums(const int32_t *__restrict__ a, int64_t count, int32_t *__restrict__ sums) {
  __shared__ int32_t red[256];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 16;
  int32_t s = 0;
  for (int q = 0; q < 16; ++q)
    if (base + q < count) s += a[base + q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void k_scan_sums(int32_t *sums, int nblocks) {  // exclusive, in place, one workgroup
  __shared__ int32_t part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
  int32_t s = 0;
  for (int b = b0; b < b1; ++b) s += sums[b];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int t = 0; t < 1024; ++t) {
      const int32_t v = part[t];
      part[t] = run;
      run += v;
    }
  }
  __syncthreads();
  int32_t run = part[threadIdx.x];
  for (int b = b0; b < b1; ++b) {
    const int32_t v = sums[b];
    sums[b] = run;
    run += v;
  }
}
__global__ __launch_bounds__(256) void k_scan_apply(int32_t *__restrict__ a, int64_t count, const int32_t *__restrict__ sums) {
  __shared__ int32_t part[256];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 16;
  int32_t v[16];
  int32_t s = 0;
  for (int q = 0; q < 16; ++q) {
    v[q] = base + q < count ? a[base + q] : 0;
    s += v[q];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = sums[blockIdx.x];
    for (int t = 0; t < 256; ++t) {
      const int32_t x = part[t];
      part[t] = run;
      run += x;
    }
  }
  __syncthreads();
  int32_t run = part[threadIdx.x];
  for (int q = 0; q < 16; ++q) {
    run += v[q];
    if (base + q < count) a[base + q] = run;
  }
}

// ---- the permuted CSR (slq.hip: csr_create_body, "permuted CSR"): stored row i = caller row perm[i], its columns renumbered by inv
// and sorted (equal columns keep the caller's order); rp2 = the stored row pointers (built on the host: a running sum over perm) ----
template <typename F>
__global__ __launch_bounds__(256) void k_permute_csr(int n, const int32_t *__restrict__ rp0, const int32_t *__restrict__ ci0, const F *__restrict__ va0,
                                                     const int32_t *__restrict__ perm, const int32_t *__restrict__ inv, const int32_t *__restrict__ rp2,
                                                     int32_t *ci2, F *va2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = perm[i];
  const int w0 = rp2[i];
  int len = 0;
  for (int q = rp0[o]; q < rp0[o + 1]; ++q) {
    const int32_t c = inv[ci0[q]];
    const F v = va0[q];
    int j = w0 + len;
    while (j > w0 && ci2[j - 1] > c) {
      ci2[j] = ci2[j - 1];
      va2[j] = va2[j - 1];
      --j;
    }
    ci2[j] = c;
    va2[j] = v;
    ++len;
  }
}

// ---- gathers that reach further than 4096 stored rows (slq.hip: "far count") ----
__global__ __launch_bounds__(256) void k_far_count(int n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, unsigned long long *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int f = 0;
  if (i < n)
    for (int q = rp[i]; q < rp[i + 1]; ++q) {
      const int d = ci[q] - i;
      f += (d > 4096 || d < -4096);
    }
  for (int o = 32; o > 0; o >>= 1) f += __shfl_down(f, o);
  if ((threadIdx.x & 63) == 0 && f) atomicAdd(out, (unsigned long long)f);
}

// ---- exactly symmetric? (slq.hip: build_symmetric_upper) rows sorted without duplicates, every off-diagonal entry mirrored with an
// equal value; cnt[i + 1] = entries of row i on or above the diagonal (cnt[0] = 0 by the caller), *bad raised otherwise ----
template <typename F>
__global__ __launch_bounds__(256) void k_sym_count(int n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const F *__restrict__ va, int32_t *cnt,
                                                   int *bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int up = 0;
  bool b = false;
  for (int q = rp[i]; q < rp[i + 1] && !b; ++q) {
    const int j = ci[q];
    if (q > rp[i] && ci[q - 1] >= j) { b = true; break; }
    up += j >= i;
    if (j == i) continue;
    int lo = rp[j], hi = rp[j + 1];
    const int end = hi;
    while (lo < hi) {  // lower bound of i in row j
      const int m = (lo + hi) >> 1;
      if (ci[m] < i) lo = m + 1;
      else hi = m;
    }
    if (lo == end || ci[lo] != i || !(va[lo] == va[q])) b = true;
  }
  cnt[i + 1] = up;
  if (b) *bad = 1;
}
template <typename F>
__global__ __launch_bounds__(256) void k_upper_fill(int n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const F *__restrict__ va,
                                                    const int32_t *__restrict__ urp, int32_t *__restrict__ uci, F *__restrict__ uva) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int w = urp[i];
  for (int q = rp[i]; q < rp[i + 1]; ++q) {
    const int j = ci[q];
    if (j < i) continue;
    uci[w] = j;
    uva[w] = j == i ? va[q] : (F)2 * va[q];
    ++w;
  }
}

// ---- per tile: the ascending list of the distinct indices of its rows and their columns (slq.hip: build_tile_meta), at a fixed
// stride of CAP words; its length in D[t + 1] (D[0] = 0 by the caller); the size of its record in units of 16 bytes, plain and
// with every row padded to whole chunks of four entries (build_ring_stream), in units[t + 1] / units_pad[t + 1].
// flags[0]: some list outgrew CAP; flags[1]: some padded record outgrows its slot (pad_limit bytes); flags[2]: the longest list ----
template <int CAP>
__global__ __launch_bounds__(64) void k_tile_lists(int ntiles, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const int32_t *__restrict__ tile_row,
                                                   int32_t *__restrict__ lists, int32_t *__restrict__ D, int32_t *__restrict__ units, int32_t *__restrict__ units_pad,
                                                   int head_bytes, int esz, int pad_limit, int *flags) {
  __shared__ int32_t lds[64 * CAP];
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= ntiles) return;
  int32_t *u = lds + threadIdx.x * CAP;
  const int r0 = tile_row[t], r1 = tile_row[t + 1];
  int cnt = 0;
  bool over = false;
  for (int r = r0; r < r1 && cnt < CAP; ++r) u[cnt++] = r;  // (consecutive: sorted already)
  if (r1 - r0 > CAP) over = true;
  int padded = 0;
  for (int r = r0; r < r1; ++r) {
    const int len = rp[r + 1] - rp[r];
    padded += max(4, (len + 3) / 4 * 4);
    for (int q = rp[r]; q < rp[r + 1]; ++q) {
      const int32_t c = ci[q];
      int lo = 0, hi = cnt;
      while (lo < hi) {
        const int m = (lo + hi) >> 1;
        if (u[m] < c) lo = m + 1;
        else hi = m;
      }
      if (lo < cnt && u[lo] == c) continue;
      if (cnt == CAP) { over = true; continue; }
      for (int k = cnt; k > lo; --k) u[k] = u[k - 1];
      u[lo] = c;
      ++cnt;
    }
  }
  for (int k = 0; k < cnt; ++k) lists[(int64_t)t * CAP + k] = u[k];
  D[t + 1] = cnt;
  const int nz = rp[r1] - rp[r0];
  const int nzp = (nz + 3) / 4 * 4;
  units[t + 1] = (head_bytes + nzp * 4 + nzp * esz + 15) / 16;
  units_pad[t + 1] = (head_bytes + padded * 4 + padded * esz + 15) / 16;
  if (over) atomicOr(&flags[0], 1);
  if (head_bytes + padded * (4 + esz) > pad_limit) atomicOr(&flags[1], 1);
  atomicMax(&flags[2], cnt);
}

// ---- a tile's descriptor (64 R words) and record (slq.hip: build_ring_stream; layouts in slq_common.hpp / slq_ring.hpp): one
// wavefront per tile, every word of both written exactly once. off[t] = the record's start in units of 16 bytes ----
template <typename F, int CAP>
__global__ __launch_bounds__(256) void k_tile_stream(int ntiles, int R, int pad, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const F *__restrict__ va,
                                                     const int32_t *__restrict__ tile_row, const int32_t *__restrict__ lists, const int32_t *__restrict__ D,
                                                     const int32_t *__restrict__ off, int32_t *__restrict__ desc, char *__restrict__ rec) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const int32_t *u = lists + (int64_t)t * CAP;
  const int r0 = tile_row[t], rows = tile_row[t + 1] - r0, p0 = rp[r0];
  const int Dt = D[t + 1] - D[t];
  const int dw = 64 * R, head_bytes = kRecHeadBytes * R, valoff_w = 16 * R - 1, self_w = 16 * R;
  const int64_t o16 = off[t];
  const int bytes = (off[t + 1] - off[t]) * 16;
  auto find = [&](int32_t c) {
    int lo = 0, hi = Dt;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (u[m] < c) lo = m + 1;
      else hi = m;
    }
    return lo;
  };
  // entries of the record: rows one after the other (padded: each row to max(4, whole chunks of four) entries)
  int nz;
  int my_start = 0;  // lane i < rows: where row i's entries start (padded records)
  if (pad) {
    // rows <= 14 R <= 56 < 64: one lane per row, a prefix sum across the wavefront
    const int len = lane < rows ? rp[r0 + lane + 1] - rp[r0 + lane] : 0;
    const int pc = lane < rows ? max(4, (len + 3) / 4 * 4) : 0;
    int incl = pc;
    for (int o = 1; o < 64; o <<= 1) {
      const int x = __shfl_up(incl, o);
      if (lane >= o) incl += x;
    }
    my_start = incl - pc;
    nz = __shfl(incl, 63);
  } else {
    nz = rp[r0 + rows] - p0;
  }
  const int nzp = (nz + 3) / 4 * 4, valoff = head_bytes + nzp * 4;
  int32_t *head = (int32_t *)(rec + o16 * 16);
  int32_t *lc_out = (int32_t *)(rec + o16 * 16 + head_bytes);
  F *va_out = (F *)(rec + o16 * 16 + valoff);
  // header words
  for (int w = lane; w < head_bytes / 4; w += 64) {
    int32_t x = 0;
    if (w <= rows) x = pad ? (w < rows ? my_start : nz) : rp[r0 + w] - p0;  // (rows < 64: w is this lane's own row here)
    else if (w == valoff_w) x = valoff;
    else if (w >= self_w && w < self_w + rows) x = find(r0 + (w - self_w));
    head[w] = x;
  }
  // entries
  if (!pad) {
    for (int e = lane; e < nzp; e += 64) {
      lc_out[e] = e < nz ? find(ci[p0 + e]) : 0;
      va_out[e] = e < nz ? va[p0 + e] : (F)0;
    }
  } else if (lane < rows) {
    const int q0 = rp[r0 + lane], cnt = rp[r0 + lane + 1] - q0, pc = max(4, (cnt + 3) / 4 * 4);
    const int self = find(r0 + lane);
    for (int q = 0; q < pc; ++q) {
      lc_out[my_start + q] = q < cnt ? find(ci[q0 + q]) : self;
      va_out[my_start + q] = q < cnt ? va[q0 + q] : (F)0;
    }
  }
  // what is left of the record behind the values (its size is a whole number of 16-byte units already: nothing), then the descriptor
  (void)bytes;
  int32_t *d = desc + (int64_t)t * dw;
  const int nd = (Dt + R - 1) / R;
  for (int w = lane; w < dw; w += 64) {
    const int b = w / 64, k = w % 64;
    int32_t x = 0;
    if (b == 0 && k < kDescList) {
      if (k == kDescCols) x = Dt;
      else if (k == kDescRecOff) x = (int32_t)o16;
      else if (k == kDescRecChunks) x = (bytes + 1023) / 1024;
      else if (k == kDescRow0) x = r0;
      else if (k == kDescRows) x = rows;
    } else if (k >= kDescList) {
      const int p = k - kDescList;
      if (R == 1) {  // de-interleaved: even lines, then odd ones (slq_common.hpp: ring1_list_pos)
        const int c = p < kRing1ListHalf ? 2 * p : 2 * (p - kRing1ListHalf) + 1;
        if (p < 2 * kRing1ListHalf && c < Dt) x = u[c];
      } else {
        const int c = p * R + b;
        if (c < nd * R) x = u[min(c, Dt - 1)];
      }
    }
    d[w] = x;
  }
}

}  // namespace slqb
