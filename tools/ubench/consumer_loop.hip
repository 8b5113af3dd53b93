// The matrix-wave inner loop of spconv_os_lc_kernel in isolation: 4 waves per workgroup (one per SIMD), per step 20
// ds_read_b128 (4 A fragments + 16 B fragments of a 16 KB tile) and 48 v_mfma_f32_16x16x32_bf16, reads one batch ahead.
// Variants: MFMAs only / reads only / both / both + one s_barrier per step with 8 more waves that only wait at it.
// hipcc --offload-arch=gfx950 -O3 -o consumer_loop consumer_loop.hip && ./consumer_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

template <bool READS, bool MFMAS, bool BARRIER, int NWAVES, int DMA = 0>
__global__ __launch_bounds__(NWAVES * 64) void k(float *out, int steps, const u32x4 *src = nullptr) {
  __shared__ u32x4 Al[4][1024];
  __shared__ u32x4 Wl[4][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  for (int i = tid; i < 4096; i += NWAVES * 64) ((u32x4 *)Al)[i] = ((u32x4 *)Wl)[i] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  if (wave >= 4) {
    if (BARRIER)
      for (int s = 0; s < steps; ++s) {
        if (DMA > 0) {
          // DMA pieces of 1 KB from an L2-resident buffer into the ring (what the loader waves of the conv kernel do)
          const int lw = wave - 4;
#pragma unroll
          for (int i = 0; i < DMA; ++i) {
            const unsigned piece = (unsigned)((s * (NWAVES - 4) + lw) * DMA + i);
            const u32x4 *p = src + (size_t)((piece * 64u + blockIdx.x * 4096u) & 0x1ffffu) + lane;     // 2 MB window
            u32x4 *dst = (i & 1) ? &Wl[s & 3][((lw * DMA + i) >> 1 & 15) * 64] : &Al[s & 3][((lw * DMA + i) >> 1 & 15) * 64];
            __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
          }
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
        }
        asm volatile("s_barrier" ::: "memory");
      }
    return;
  }
  auto swz = [](int nn) { return ((nn >> 2) & 1) | (((nn >> 1) & 1) << 2); };
  const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&Al[0][0] + (unsigned)((wave * 32 + n) * 8) * 16u;
  const unsigned a0 = a_lds + (unsigned)((g * 2) ^ swz(n)) * 16u, a1 = a_lds + (unsigned)((g * 2 + 1) ^ swz(n)) * 16u;
  const unsigned w0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&Wl[0][0] + (unsigned)lane * 16u;
  f32x4 acc[2][8];
  for (int i = 0; i < 16; ++i) acc[i >> 3][i & 7] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 af[2][2][2], bq[2][4];
  for (int i = 0; i < 8; ++i) af[i >> 2][(i >> 1) & 1][i & 1] = bq[i >> 2][i & 3] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#define RD(dst, addr, off)                                                                              \
  do {                                                                                                  \
    if (READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));         \
  } while (0)
#define RD_A(buf, t)                                              \
  do {                                                            \
    const unsigned o_ = (unsigned)((t) & 3) * 16384u;             \
    RD(af[buf][0][0], a0 + o_, 0);                                \
    RD(af[buf][0][1], a1 + o_, 0);                                \
    RD(af[buf][1][0], a0 + o_, 2048);                             \
    RD(af[buf][1][1], a1 + o_, 2048);                             \
  } while (0)
#define RD_B(i, slot, t)                                          \
  do {                                                            \
    const unsigned w_ = w0 + (unsigned)((t) & 3) * 16384u;        \
    RD(bq[slot][0], w_, ((i) * 4 + 0) * 1024);                    \
    RD(bq[slot][1], w_, ((i) * 4 + 1) * 1024);                    \
    RD(bq[slot][2], w_, ((i) * 4 + 2) * 1024);                    \
    RD(bq[slot][3], w_, ((i) * 4 + 3) * 1024);                    \
  } while (0)
#define WAIT_B(cnt, slot)                                                                                                    \
  do {                                                                                                                       \
    if (READS) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(bq[slot][0]), "+v"(bq[slot][1]), "+v"(bq[slot][2]), "+v"(bq[slot][3])); \
  } while (0)
#define WAIT_AB(cnt, buf, slot)                                                                                          \
  do {                                                                                                                   \
    if (READS)                                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                         \
                   : "+v"(af[buf][0][0]), "+v"(af[buf][0][1]), "+v"(af[buf][1][0]), "+v"(af[buf][1][1]), "+v"(bq[slot][0]), \
                     "+v"(bq[slot][1]), "+v"(bq[slot][2]), "+v"(bq[slot][3]));                                           \
  } while (0)
#define BATCH(i, buf, slot)                                                                   \
  do {                                                                                        \
    if (MFMAS) {                                                                              \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                      \
        acc[rt][(i) * 2] = MFMA(af[buf][rt][1], bq[slot][0], acc[rt][(i) * 2]);               \
        acc[rt][(i) * 2 + 1] = MFMA(af[buf][rt][1], bq[slot][2], acc[rt][(i) * 2 + 1]);       \
      }                                                                                       \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                      \
        acc[rt][(i) * 2] = MFMA(af[buf][rt][0], bq[slot][1], acc[rt][(i) * 2]);               \
        acc[rt][(i) * 2 + 1] = MFMA(af[buf][rt][0], bq[slot][3], acc[rt][(i) * 2 + 1]);       \
      }                                                                                       \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                      \
        acc[rt][(i) * 2] = MFMA(af[buf][rt][0], bq[slot][0], acc[rt][(i) * 2]);               \
        acc[rt][(i) * 2 + 1] = MFMA(af[buf][rt][0], bq[slot][2], acc[rt][(i) * 2 + 1]);       \
      }                                                                                       \
    }                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#define STEP(s, buf)                                              \
  do {                                                            \
    WAIT_AB(4, buf, 0);                                           \
    BATCH(0, buf, 0);                                             \
    RD_B(2, 0, s);                                                \
    WAIT_B(4, 1);                                                 \
    BATCH(1, buf, 1);                                             \
    RD_B(3, 1, s);                                                \
    WAIT_B(4, 0);                                                 \
    BATCH(2, buf, 0);                                             \
    RD_A((buf) ^ 1, (s) + 1);                                     \
    RD_B(0, 0, (s) + 1);                                          \
    WAIT_B(8, 1);                                                 \
    BATCH(3, buf, 1);                                             \
    RD_B(1, 1, (s) + 1);                                          \
    if (BARRIER) asm volatile("s_barrier" ::: "memory");          \
  } while (0)
  RD_A(0, 0);
  RD_B(0, 0, 0);
  RD_B(1, 1, 0);
  for (int s = 0; s < steps; s += 2) {
    STEP(s, 0);
    STEP(s + 1, 1);
  }
  if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f32x4 t = acc[0][0];
  for (int i = 1; i < 16; ++i) t += acc[i >> 3][i & 7];
  t[0] += __uint_as_float(af[0][0][0][0] ^ bq[0][0][0] ^ af[1][1][1][1] ^ bq[1][3][2]);
  out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
}


// Same work, reads interleaved one by one behind the first MFMAs of the PREVIOUS batch (one batch ahead, never clustered)
template <int NWAVES, int DMA>
__global__ __launch_bounds__(NWAVES * 64) void k2(float *out, int steps, const u32x4 *src) {
  __shared__ u32x4 Al[4][1024];
  __shared__ u32x4 Wl[4][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  for (int i = tid; i < 4096; i += NWAVES * 64) ((u32x4 *)Al)[i] = ((u32x4 *)Wl)[i] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  if (wave >= 4) {
    for (int s = 0; s < steps; ++s) {
      if (DMA > 0) {
        const int lw = wave - 4;
#pragma unroll
        for (int i = 0; i < DMA; ++i) {
          const unsigned piece = (unsigned)((s * (NWAVES - 4) + lw) * DMA + i);
          const u32x4 *p = src + (size_t)((piece * 64u + blockIdx.x * 4096u) & 0x1ffffu) + lane;
          u32x4 *dst = (i & 1) ? &Wl[s & 3][((lw * DMA + i) >> 1 & 15) * 64] : &Al[s & 3][((lw * DMA + i) >> 1 & 15) * 64];
          __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
      }
      asm volatile("s_barrier" ::: "memory");
    }
    return;
  }
  auto swz = [](int nn) { return ((nn >> 2) & 1) | (((nn >> 1) & 1) << 2); };
  const unsigned a_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&Al[0][0] + (unsigned)((wave * 32 + n) * 8) * 16u;
  const unsigned a0 = a_lds + (unsigned)((g * 2) ^ swz(n)) * 16u, a1 = a_lds + (unsigned)((g * 2 + 1) ^ swz(n)) * 16u;
  const unsigned w0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&Wl[0][0] + (unsigned)lane * 16u;
  f32x4 acc[2][8];
  for (int i = 0; i < 16; ++i) acc[i >> 3][i & 7] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 af[2][2][2], bq[2][4];
#define R2(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define SB() __builtin_amdgcn_sched_barrier(0)
#define M(rt, c, ap, br, buf, slot) acc[rt][c] = MFMA(af[buf][rt][ap], bq[slot][br], acc[rt][c])
  // batch i of the step in stage t: 12 MFMAs; RA..RH = statements issued behind MFMAs 0..7
#define BATCH2(i, buf, slot, RA, RB, RC, RD_, RE, RF, RG, RH)   \
  do {                                                          \
    M(0, (i) * 2, 1, 0, buf, slot); SB(); RA; SB();             \
    M(0, (i) * 2 + 1, 1, 2, buf, slot); SB(); RB; SB();         \
    M(1, (i) * 2, 1, 0, buf, slot); SB(); RC; SB();             \
    M(1, (i) * 2 + 1, 1, 2, buf, slot); SB(); RD_; SB();        \
    M(0, (i) * 2, 0, 1, buf, slot); SB(); RE; SB();             \
    M(0, (i) * 2 + 1, 0, 3, buf, slot); SB(); RF; SB();         \
    M(1, (i) * 2, 0, 1, buf, slot); SB(); RG; SB();             \
    M(1, (i) * 2 + 1, 0, 3, buf, slot); SB(); RH; SB();         \
    M(0, (i) * 2, 0, 0, buf, slot);                             \
    M(0, (i) * 2 + 1, 0, 2, buf, slot);                         \
    M(1, (i) * 2, 0, 0, buf, slot);                             \
    M(1, (i) * 2 + 1, 0, 2, buf, slot);                         \
    SB();                                                       \
  } while (0)
#define WAITALL(buf)                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(af[buf][0][0]), "+v"(af[buf][0][1]), "+v"(af[buf][1][0]), "+v"(af[buf][1][1]), "+v"(bq[0][0]),   \
                 "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]))
#define NOP_ do { } while (0)
#define STEP2(s, buf)                                                                                                  \
  do {                                                                                                                 \
    const unsigned wt = w0 + (unsigned)((s) & 3) * 16384u, wn = w0 + (unsigned)(((s) + 1) & 3) * 16384u;               \
    const unsigned an0 = a0 + (unsigned)(((s) + 1) & 3) * 16384u, an1 = a1 + (unsigned)(((s) + 1) & 3) * 16384u;       \
    WAITALL(buf);                                                                                                      \
    BATCH2(0, buf, 0, R2(bq[1][0], wt, 4 * 1024), R2(bq[1][1], wt, 5 * 1024), R2(bq[1][2], wt, 6 * 1024),              \
           R2(bq[1][3], wt, 7 * 1024), NOP_, NOP_, NOP_, NOP_);                                                        \
    WAITALL(buf);                                                                                                      \
    BATCH2(1, buf, 1, R2(bq[0][0], wt, 8 * 1024), R2(bq[0][1], wt, 9 * 1024), R2(bq[0][2], wt, 10 * 1024),             \
           R2(bq[0][3], wt, 11 * 1024), NOP_, NOP_, NOP_, NOP_);                                                       \
    WAITALL(buf);                                                                                                      \
    BATCH2(2, buf, 0, R2(bq[1][0], wt, 12 * 1024), R2(bq[1][1], wt, 13 * 1024), R2(bq[1][2], wt, 14 * 1024),           \
           R2(bq[1][3], wt, 15 * 1024), R2(af[(buf) ^ 1][0][0], an0, 0), R2(af[(buf) ^ 1][0][1], an1, 0),              \
           R2(af[(buf) ^ 1][1][0], an0, 2048), R2(af[(buf) ^ 1][1][1], an1, 2048));                                    \
    WAITALL(buf);                                                                                                      \
    BATCH2(3, buf, 1, R2(bq[0][0], wn, 0 * 1024), R2(bq[0][1], wn, 1 * 1024), R2(bq[0][2], wn, 2 * 1024),              \
           R2(bq[0][3], wn, 3 * 1024), NOP_, NOP_, NOP_, NOP_);                                                        \
    asm volatile("s_barrier" ::: "memory");                                                                            \
  } while (0)
  R2(af[0][0][0], a0, 0);
  R2(af[0][0][1], a1, 0);
  R2(af[0][1][0], a0, 2048);
  R2(af[0][1][1], a1, 2048);
  R2(bq[0][0], w0, 0);
  R2(bq[0][1], w0, 1024);
  R2(bq[0][2], w0, 2048);
  R2(bq[0][3], w0, 3072);
  for (int s = 0; s < steps; s += 2) {
    STEP2(s, 0);
    STEP2(s + 1, 1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f32x4 t = acc[0][0];
  for (int i = 1; i < 16; ++i) t += acc[i >> 3][i & 7];
  t[0] += __uint_as_float(af[0][0][0][0] ^ bq[0][0][0] ^ af[1][1][1][1] ^ bq[1][3][2]);
  out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
}

template <int NWAVES, int DMA>
void run2(const char *name, float *out, const u32x4 *src) {
  const int steps = 2000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<NWAVES, DMA>), dim3(blocks), dim3(NWAVES * 64), 0, 0, out, 20, src);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k2<NWAVES, DMA>), dim3(blocks), dim3(NWAVES * 64), 0, 0, out, steps, src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.1f ns per step = %5.0f clocks @2.4 GHz (48 MFMAs = 768)\n", name, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}

template <bool READS, bool MFMAS, bool BARRIER, int NWAVES, int DMA = 0>
void run(const char *name, float *out, const u32x4 *src = nullptr) {
  const int steps = 2000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<READS, MFMAS, BARRIER, NWAVES, DMA>), dim3(blocks), dim3(NWAVES * 64), 0, 0, out, 20, src);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<READS, MFMAS, BARRIER, NWAVES, DMA>), dim3(blocks), dim3(NWAVES * 64), 0, 0, out, steps, src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.1f ns per step = %5.0f clocks @2.4 GHz (48 MFMAs = 768)\n", name, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 1024 * 4);
  run<false, true, false, 4>("MFMAs only, 4 waves", out);
  run<true, false, false, 4>("reads only, 4 waves", out);
  run<true, true, false, 4>("reads + MFMAs, 4 waves", out);
  run<true, true, true, 4>("reads + MFMAs + barrier, 4 waves", out);
  run<true, true, true, 12>("reads + MFMAs + barrier, 4 + 8 waiting waves", out);
  run<false, true, true, 12>("MFMAs + barrier, 4 + 8 waiting waves", out);
  u32x4 *src;
  hipMalloc(&src, 4 << 20);
  hipMemset(src, 0x3f, 4 << 20);
  run<true, true, true, 12, 4>("reads + MFMAs + barrier, 4 + 8 waves x 4 DMA pieces", out, src);
  run<true, true, true, 12, 2>("reads + MFMAs + barrier, 4 + 8 waves x 2 DMA pieces", out, src);
  run<true, true, true, 8, 8>("reads + MFMAs + barrier, 4 + 4 waves x 8 DMA pieces", out, src);
  run2<12, 0>("interleaved reads + MFMAs + barrier, 4 + 8 waiting waves", out, src);
  run2<12, 4>("interleaved reads + MFMAs + barrier, 4 + 8 waves x 4 DMA", out, src);
  run<false, true, true, 12, 4>("MFMAs + barrier, 4 + 8 waves x 4 DMA pieces", out, src);
  run<true, false, true, 12, 4>("reads + barrier, 4 + 8 waves x 4 DMA pieces", out, src);
  run<false, false, true, 12, 4>("barrier only, 4 + 8 waves x 4 DMA pieces", out, src);
  return 0;
}
