// Error reporting, device queries and the device-wide scan used by the rulebook / voxeliser.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace df3d {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------- fp16 operand splits: range flags
// One `static __device__` flag per translation unit (common.h); their host symbols are collected here by static
// constructors (a plain array: no initialisation order to get wrong) and resolved to device addresses at the first poll.
static const void *g_ovf_sym[64];
static const char *g_ovf_tu[64];
static constexpr int OVF_MAXDEV = 16;
static unsigned *g_ovf_dev[OVF_MAXDEV][64];       // device addresses of the flags, per device (they differ between GPUs)
static unsigned **g_ovf_table[OVF_MAXDEV];         // the same addresses as a device array (df3d_split_overflow_collect)
static int g_ovf_table_n[OVF_MAXDEV];
static int g_ovf_n = 0;

// the flags' addresses on the CURRENT device (resolved once per device); -> device index or -1
static int ovf_resolve() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OVF_MAXDEV) return -1;
  for (int i = 0; i < g_ovf_n; ++i) {
    if (g_ovf_dev[dev][i]) continue;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, g_ovf_sym[i]) != hipSuccess || !p) {
      (void)hipGetLastError();
      continue;                                     // (no device code of that unit was loaded)
    }
    g_ovf_dev[dev][i] = (unsigned *)p;
  }
  return dev;
}

__global__ void ovf_collect_kernel(unsigned *const *flags, int n, unsigned *out, int reset) {
  unsigned m = 0;
  for (int i = threadIdx.x; i < n; i += 64) {
    unsigned *f = flags[i];
    if (f && *f) {
      m |= 1u << (i & 31);
      if (reset) *f = 0u;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
  if (threadIdx.x == 0) *out = m;
}
void split_overflow_register(const void *symbol, const char *tu) {
  if (g_ovf_n < 64) g_ovf_sym[g_ovf_n] = symbol, g_ovf_tu[g_ovf_n] = tu, ++g_ovf_n;
}

// ---------------------------------------------------------------------------- scan
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

struct LoadU32 {
  const uint32_t *p;
  __device__ uint32_t operator()(size_t i) const { return p[i]; }
};
struct LoadPopc {
  const unsigned long long *p;
  __device__ uint32_t operator()(size_t i) const { return (uint32_t)__popcll(p[i]); }
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan inside one block of SCAN_TILE items; returns block total in *total (all threads)
__device__ __forceinline__ void block_excl_scan(uint32_t (&x)[SCAN_ITEMS], uint32_t &total,
                                                uint32_t *lds /*[SCAN_THREADS/64 + 1]*/) {
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) s += x[i];
  uint32_t inc = wave_incl_scan(s, lane);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    uint32_t v = lds[w];
    if (w < wave) woff += v;
    tot += v;
  }
  uint32_t run = woff + inc - s;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint32_t v = x[i];
    x[i] = run;
    run += v;
  }
  total = tot;
  __syncthreads();
}

template <typename Loader>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(Loader in, uint32_t *out, size_t n,
                                                                  uint32_t *sums, uint32_t *total_if_single) {
  __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t x[SCAN_ITEMS];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) x[i] = (base + i < n) ? in(base + i) : 0u;
  uint32_t tot;
  block_excl_scan(x, tot, lds);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) out[base + i] = x[i];
  if (threadIdx.x == 0) {
    sums[blockIdx.x] = tot;
    if (total_if_single) *total_if_single = tot;      // one tile: the scan is complete after this launch
  }
}

// second (and last) launch of a multi-tile scan: every block sums the totals of the tiles in front of it (at most a
// few thousand values out of L2) and adds the offset to its tile -- no separate launch for the scan of the tile sums
__global__ __launch_bounds__(SCAN_THREADS) void scan_offset_kernel(uint32_t *out, size_t n, const uint32_t *sums, size_t nb,
                                                                   uint32_t *total) {
  __shared__ uint32_t lds[SCAN_THREADS / 64];
  __shared__ uint32_t s_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool last = blockIdx.x + 1 == nb;
  const size_t upto = last ? nb : blockIdx.x;          // the last block also produces the grand total
  uint32_t s = 0, mine = 0;
  for (size_t i = tid; i < upto; i += SCAN_THREADS) {
    const uint32_t v = sums[i];
    if (i < blockIdx.x) mine += v;
    s += v;
  }
  // two reductions (offset of this tile, grand total) share the shuffles
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mine += __shfl_xor(mine, d, 64);
    s += __shfl_xor(s, d, 64);
  }
  __shared__ uint32_t lds2[SCAN_THREADS / 64];
  if (lane == 0) {
    lds[wave] = mine;
    lds2[wave] = s;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t a = 0, b = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
      a += lds[w];
      b += lds2[w];
    }
    s_off = a;
    if (last && total) *total = b;
  }
  __syncthreads();
  const uint32_t add = s_off;
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) out[base + i] += add;
}

size_t scan_scratch_bytes(size_t n) { return align_up((n / SCAN_TILE + 2) * sizeof(uint32_t), 256); }

template <typename Loader>
static int scan_impl(Loader in, uint32_t *out, size_t n, uint32_t *total, void *scratch, size_t scratch_bytes,
                     hipStream_t stream) {
  if (n == 0) {
    if (total) DF3D_HIP(hipMemsetAsync(total, 0, sizeof(uint32_t), stream));
    return DF3D_OK;
  }
  if (scratch_bytes < scan_scratch_bytes(n)) {
    set_error("scan scratch too small");
    return DF3D_ENOMEM;
  }
  uint32_t *sums = (uint32_t *)scratch;
  size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(scan_tiles_kernel<Loader>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, out, n, sums,
                     nb == 1 ? total : (uint32_t *)nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(scan_offset_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, out, n, sums, nb, total);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total, void *scratch,
                       size_t scratch_bytes, hipStream_t stream) {
  return scan_impl(LoadU32{in}, out, n, total, scratch, scratch_bytes, stream);
}
int exclusive_scan_popc64(const unsigned long long *words, uint32_t *out, size_t n, uint32_t *total,
                          void *scratch, size_t scratch_bytes, hipStream_t stream) {
  return scan_impl(LoadPopc{words}, out, n, total, scratch, scratch_bytes, stream);
}

}  // namespace df3d

extern "C" {

int df3d_version(void) { return 100; }

int df3d_split_overflow(int reset, char *where, int where_len) {
  // synchronous on purpose (a poll at a frame / test boundary): waits for the CURRENT device
  unsigned any = 0;
  if (where && where_len > 0) where[0] = 0;
  if (hipDeviceSynchronize() != hipSuccess) {
    df3d::set_error("df3d_split_overflow: hipDeviceSynchronize failed");
    return DF3D_EHIP;
  }
  const int dev = df3d::ovf_resolve();
  if (dev < 0) {
    df3d::set_error("df3d_split_overflow: no current device");
    return DF3D_EHIP;
  }
  for (int i = 0; i < df3d::g_ovf_n; ++i) {
    unsigned *f = df3d::g_ovf_dev[dev][i];
    if (!f) continue;
    unsigned v = 0;
    DF3D_HIP(hipMemcpy(&v, f, sizeof(v), hipMemcpyDeviceToHost));
    if (v) {
      any |= v;
      if (where && where_len > 0) {
        size_t used = strlen(where);
        snprintf(where + used, (size_t)where_len - used, "%s%s", used ? "," : "", df3d::g_ovf_tu[i]);
      }
      if (reset) DF3D_HIP(hipMemset(f, 0, sizeof(v)));
    }
  }
  return any ? 1 : 0;
}

// The same poll WITHOUT a host wait (round 6): one tiny launch on `stream` ORs every unit's flag into *out (a device word of the
// caller; bit i & 31 = unit i of df3d_split_overflow_units) and clears the flags when `reset`.  The caller reads the word with a
// device -> host copy it makes anyway (a detector's box counts, a trainer's logged losses): the flag then covers every kernel
// queued on `stream` before this call.
int df3d_split_overflow_collect(uint32_t *out, int reset, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(out, "split_overflow_collect: null output");
  const int dev = df3d::ovf_resolve();
  if (dev < 0) {
    df3d::set_error("df3d_split_overflow_collect: no current device");
    return DF3D_EHIP;
  }
  // (a unit's flag has an address once its code object is loaded -- HIP loads them lazily, at the unit's first launch: the
  // device-side table is refreshed whenever more of them have resolved)
  int resolved = 0;
  for (int i = 0; i < df3d::g_ovf_n; ++i) resolved += df3d::g_ovf_dev[dev][i] != nullptr;
  if (!df3d::g_ovf_table[dev] || df3d::g_ovf_table_n[dev] != resolved) {
    if (!df3d::g_ovf_table[dev]) DF3D_HIP(hipMalloc((void **)&df3d::g_ovf_table[dev], sizeof(unsigned *) * 64));
    DF3D_HIP(hipMemcpyAsync(df3d::g_ovf_table[dev], df3d::g_ovf_dev[dev], sizeof(unsigned *) * 64, hipMemcpyHostToDevice, stream));
    DF3D_HIP(hipStreamSynchronize(stream));         // (the host array may change before an asynchronous copy has read it; rare)
    df3d::g_ovf_table_n[dev] = resolved;
  }
  hipLaunchKernelGGL(df3d::ovf_collect_kernel, dim3(1), dim3(64), 0, stream, df3d::g_ovf_table[dev], df3d::g_ovf_n, out, reset);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// names of the units behind the bits of df3d_split_overflow_collect, comma separated in registration order
int df3d_split_overflow_units(char *buf, int buflen) {
  if (!buf || buflen <= 0) return df3d::g_ovf_n;
  buf[0] = 0;
  for (int i = 0; i < df3d::g_ovf_n; ++i) {
    size_t used = strlen(buf);
    snprintf(buf + used, (size_t)buflen - used, "%s%s", used ? "," : "", df3d::g_ovf_tu[i]);
  }
  return df3d::g_ovf_n;
}
const char *df3d_last_error(void) { return df3d::g_err; }

int df3d_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int df3d_device_arch(char *buf, int buflen) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) {
    df3d::set_error("hipGetDeviceProperties failed");
    return DF3D_EHIP;
  }
  snprintf(buf, buflen, "%s", p.gcnArchName);
  return DF3D_OK;
}
}
