// VALU issue-rate microbenchmark for gfx950 (MI355X), round 2: settles how many cycles one wave64
// VALU instruction occupies a SIMD (the guide lists v_fma_f32 at 2 cycles,
// round 1 measured about 4).
//
// Method: exactly W waves resident per SIMD (grid = CUs x 4 x W single-wave workgroups, few
// registers), every wave runs the same long stream of ONE instruction over 16 INDEPENDENT
// accumulators (no instruction depends on any of the previous 15), for a few milliseconds, and
// stamps the shader cycle counter (s_memtime) and the constant 100 MHz clock (s_memrealtime) at
// both ends.  Reported per (instruction, W):
//   cycles/instr/SIMD = launch time (HIP events) x measured clock / (W * instructions per wave)  -- the issue interval
//   clock             = cycle delta / realtime delta                        -- what the chip ran at
//   wave-instr/us/SIMD from the host-side HIP events, for comparison with round 1
// W = 1 shows the single-wave issue limit, W >= 2 the SIMD's.
//
// Build:  hipcc -O3 --offload-arch=gfx950 tools/valu_rates2.hip -o tools/valu_rates2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                       \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) {                                                                            \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);           \
      exit(1);                                                                                         \
    }                                                                                                  \
  } while (0)

constexpr int NACC = 16;    // independent accumulators per lane
constexpr int INNER = 16;   // passes over the accumulators per loop trip (256 instructions per trip)

enum Op {
  FMA_F32, FMAC_F32, PK_FMA_F32, FMA_F64, MAD_U64_U32, MAD_U64_U32_SGPRCARRY, MAD_I64_I32, MUL_LO_U32, MUL_HI_U32,
  ADD_U32, AND_B32, XOR_B32, LSHLREV_B32, MOV_B32, ADD3_U32, AND_OR_B32, ALIGNBIT_B32, BFE_U32, LSHRREV_B64, LSHL_ADD_U64,
  ADD_CO_ADDC, LADDER_MIX, MAD_ADD_1_1, MAD_ADD_1_3, NUM_OPS
};
static const char* op_names[NUM_OPS] = {
    "v_fma_f32 (VOP3)", "v_fmac_f32 (VOP2)", "v_pk_fma_f32", "v_fma_f64", "v_mad_u64_u32 (carry -> vcc)",
    "v_mad_u64_u32 (carry -> s[n:n+1])", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32 (VOP2)",
    "v_and_b32 (VOP2)", "v_xor_b32 (VOP2)", "v_lshlrev_b32 (VOP2)", "v_mov_b32 (VOP1)", "v_add3_u32 (VOP3)",
    "v_and_or_b32 (VOP3)", "v_alignbit_b32 (VOP3)", "v_bfe_u32 (VOP3)", "v_lshrrev_b64 (VOP3)", "v_lshl_add_u64 (VOP3)",
    "v_add_co_u32 + v_addc_co_u32 (per instr)", "ladder mix: 10 mad + 2 shift64 + 2 and + 2 add (per instr)", "v_mad_u64_u32 : v_add_u32 = 1 : 1 (per instr)",
    "v_mad_u64_u32 : v_add_u32 = 1 : 3 (per instr)"};

template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(uint32_t* out, unsigned long long* stamps, int trips) {
  uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a = tid * 2654435761u + 12345u, b = tid * 40503u + 977u;
  uint64_t acc[NACC];
  double dacc[NACC];
  float facc[NACC];
  uint32_t c32[NACC];
  // only the accumulator file the instruction under test uses is kept live (<= 64 VGPRs, so that
  // 8 waves per SIMD are resident together)
  constexpr bool use_acc = OP == MAD_U64_U32 || OP == MAD_U64_U32_SGPRCARRY || OP == MAD_I64_I32 || OP == LSHRREV_B64 ||
                           OP == LSHL_ADD_U64 || OP == ADD_CO_ADDC || OP == LADDER_MIX || OP == MAD_ADD_1_1 || OP == MAD_ADD_1_3;
  constexpr bool use_d = OP == FMA_F64 || OP == PK_FMA_F32;
  constexpr bool use_f = OP == FMA_F32 || OP == FMAC_F32;
  constexpr bool use_c = !use_acc && !use_d && !use_f || OP == LADDER_MIX || OP == MAD_ADD_1_1 || OP == MAD_ADD_1_3;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    if constexpr (use_acc) acc[i] = a + i;
    if constexpr (use_d) dacc[i] = (double)(a & 0xffff) + i;
    if constexpr (use_f) facc[i] = (float)(b & 0xff) + i;
    if constexpr (use_c) c32[i] = b + i;
  }
  double da = 1.0000001, db = 0.9999999;
  float fa = 1.0001f, fb = 0.9999f;
  uint64_t sdump = 0;
  asm volatile("" : "+v"(a), "+v"(b));
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int k = 0; k < INNER; ++k) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (OP == FMA_F32) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(facc[i]) : "v"(fa), "v"(fb));
        } else if constexpr (OP == FMAC_F32) {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(facc[i]) : "v"(fa), "v"(fb));
        } else if constexpr (OP == PK_FMA_F32) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
        } else if constexpr (OP == FMA_F64) {
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
        } else if constexpr (OP == MAD_U64_U32) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == MAD_U64_U32_SGPRCARRY) {
          asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(sdump) : "v"(a), "v"(b));
        } else if constexpr (OP == MAD_I64_I32) {
          asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == MUL_LO_U32) {
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == MUL_HI_U32) {
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == ADD_U32) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == AND_B32) {
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == XOR_B32) {
          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == LSHLREV_B32) {
          asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(c32[i]));
        } else if constexpr (OP == MOV_B32) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(c32[i]) : "v"(a));
        } else if constexpr (OP == ADD3_U32) {
          asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c32[i]) : "v"(a), "v"(b));
        } else if constexpr (OP == AND_OR_B32) {
          asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c32[i]) : "v"(a), "v"(b));
        } else if constexpr (OP == ALIGNBIT_B32) {
          asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == BFE_U32) {
          asm volatile("v_bfe_u32 %0, %0, 3, 26" : "+v"(c32[i]));
        } else if constexpr (OP == LSHRREV_B64) {
          asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[i]));
        } else if constexpr (OP == LSHL_ADD_U64) {
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % NACC]));
        } else if constexpr (OP == ADD_CO_ADDC) {
          uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
          if (i < NACC / 2) {
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
            acc[i] = ((uint64_t)hi << 32) | lo;
          }
        } else if constexpr (OP == MAD_ADD_1_1 || OP == MAD_ADD_1_3) {
          if ((OP == MAD_ADD_1_1) ? (i % 2 == 0) : (i % 4 == 0)) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
          } else {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
          }
        } else if constexpr (OP == LADDER_MIX) {
          // the instruction mix of the verification ladder's field product, 16 instructions per pass
          if (i < 10) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
          } else if (i < 12) {
            asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i]));
          } else if (i < 14) {
            asm volatile("v_and_b32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
          } else {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
          }
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    if constexpr (use_acc) s += acc[i];
    if constexpr (use_d) s += (uint64_t)dacc[i];
    if constexpr (use_f) s += (uint64_t)facc[i];
    if constexpr (use_c) s += c32[i];
  }
  out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)sdump;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int OP>
void run(int waves_per_simd, uint32_t* d_out, unsigned long long* d_st, int n_cu, double wall_mhz) {
  const int trips = 6000;   // 1.5 M instructions per wave: a few ms
  int blocks = n_cu * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 64>>>(d_out, d_st, 200);   // warm the clock
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 64>>>(d_out, d_st, trips);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(2 * blocks);
  CHECK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> cyc(blocks), mhz(blocks);
  for (int i = 0; i < blocks; ++i) {
    cyc[i] = (double)st[2 * i];
    mhz[i] = st[2 * i + 1] ? (double)st[2 * i] / (double)st[2 * i + 1] * wall_mhz : 0.0;
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(mhz.begin(), mhz.end());
  double per_pass = (OP == ADD_CO_ADDC) ? NACC : NACC;   // ADD_CO_ADDC: 8 pairs = 16 instructions per pass
  double instr_per_wave = (double)trips * INNER * per_pass;
  // issue interval from the wall time of the whole launch and the clock the waves measured (the
  // per-wave stamps alone mislead when a launch's waves are not all resident from start to end)
  double cpi = ms * 1e-3 * mhz[blocks / 2] * 1e6 / (waves_per_simd * instr_per_wave);
  (void)cyc;
  double total_wave_instr = instr_per_wave * blocks;
  printf("%-62s W=%d  cycles/instr/SIMD=%6.3f  clock=%7.1f MHz  events: %8.3f ms  %7.1f wave-instr/us/SIMD\n", op_names[OP],
         waves_per_simd, cpi, mhz[blocks / 2], ms, total_wave_instr / (n_cu * 4) / (ms * 1e3));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

template <int OP>
void sweep(uint32_t* d_out, unsigned long long* d_st, int n_cu, double wall_mhz) {
  for (int w : {1, 2, 4, 8}) run<OP>(w, d_out, d_st, n_cu, wall_mhz);
}

template <int... OPS>
void sweep_all(uint32_t* d_out, unsigned long long* d_st, int n_cu, double wall_mhz) {
  (sweep<OPS>(d_out, d_st, n_cu, wall_mhz), ...);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  int n_cu = p.multiProcessorCount, wall_khz = 0;
  CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  printf("device: %s  CUs=%d  clockRate=%d kHz  wallClockRate=%d kHz  arch=%s  accumulators=%d\n", p.name, n_cu, p.clockRate,
         wall_khz, p.gcnArchName, NACC);
  uint32_t* d_out;
  unsigned long long* d_st;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 4 * 8 * 64 * sizeof(uint32_t)));
  CHECK(hipMalloc(&d_st, (size_t)n_cu * 4 * 8 * 2 * sizeof(unsigned long long)));
  sweep_all<FMA_F32, FMAC_F32, PK_FMA_F32, FMA_F64, MAD_U64_U32, MAD_U64_U32_SGPRCARRY, MAD_I64_I32, MUL_LO_U32, MUL_HI_U32,
            ADD_U32, AND_B32, XOR_B32, LSHLREV_B32, MOV_B32, ADD3_U32, AND_OR_B32, ALIGNBIT_B32, BFE_U32, LSHRREV_B64,
            LSHL_ADD_U64, ADD_CO_ADDC, LADDER_MIX, MAD_ADD_1_1, MAD_ADD_1_3>(d_out, d_st, n_cu, wall_khz * 1e-3);
  return 0;
}
