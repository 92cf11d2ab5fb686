// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Measures sustained issue rate (wave-instructions / cycle / SIMD) of the integer and
// fp64 instructions a 256-bit modular multiply can be built from, so that the integer
// roofline used by bench.py is a measured number, not a guess.
//
// Build:  hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates
// Run:    ./tools/valu_rates            (prints one line per op and waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 8;     // independent chains per lane
constexpr int INNER  = 32;    // instructions per chain per loop trip

enum Op { MAD_U64_U32, MAD_U64_U32_DEP, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24,
          ADD_CO_PAIR, ADD3_U32, FMA_F64, FMA_F32, MAD64_PLUS_ADDC, LSHL_ADD_U64, PK_MAD_U16,
          AND_OR, CNDMASK, LSHRREV_B64, ALIGNBIT, AND_B32, ADD_U32, SUB_U32, MAD_I64_I32, BFE_U32, CNDMASK_SGPR, MOV_B32, MAD_U64_NOVCC, NUM_OPS };
static const char* op_names[NUM_OPS] = {
  "v_mad_u64_u32 (8 indep chains)", "v_mad_u64_u32 (1 dependent chain)", "v_mul_lo_u32", "v_mul_hi_u32",
  "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_add_co_u32+v_addc_co_u32 (per instr)", "v_add3_u32",
  "v_fma_f64", "v_fma_f32", "v_mad_u64_u32+v_addc_co_u32 (per pair)", "v_lshl_add_u64", "v_pk_mad_u16",
  "v_and_or_b32", "v_cndmask_b32", "v_lshrrev_b64", "v_alignbit_b32", "v_and_b32", "v_add_u32", "v_sub_u32",
  "v_mad_i64_i32", "v_bfe_u32", "v_cndmask_b32 (sgpr pair mask)", "v_mov_b32", "v_mad_u64_u32 (carry to s[..], 8 chains)" };
// instructions issued per "unit" reported


template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(uint32_t* out, unsigned long long* cyc, int trips) {
  uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a = tid * 2654435761u + 12345u, b = tid * 40503u + 977u;
  uint64_t acc[UNROLL];
  double   dacc[UNROLL];
  float    facc[UNROLL];
  uint32_t c32[UNROLL];
  for (int i = 0; i < UNROLL; ++i) { acc[i] = a + i; dacc[i] = (double)(a & 0xffff) + i; facc[i] = (float)(b & 0xff) + i; c32[i] = b + i; }
  double da = 1.0000001, db = 0.9999999; float fa = 1.0001f, fb = 0.9999f;
  uint64_t smask = 0x5555555555555555ull, sdump = 0;
  asm volatile("" : "+v"(a), "+v"(b));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int k = 0; k < INNER; ++k) {
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) {
        if constexpr (OP == MAD_U64_U32) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == MAD_U64_U32_DEP) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == MUL_LO_U32) {
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == MUL_HI_U32) {
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == MAD_U32_U24) {
          asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a), "v"(b));
        } else if constexpr (OP == MUL_HI_U32_U24) {
          asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == ADD_CO_PAIR) {
          uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
          asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
          acc[i] = ((uint64_t)hi << 32) | lo;
        } else if constexpr (OP == ADD3_U32) {
          asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c32[i]) : "v"(a), "v"(b));
        } else if constexpr (OP == FMA_F64) {
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
        } else if constexpr (OP == FMA_F32) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(facc[i]) : "v"(fa), "v"(fb));
        } else if constexpr (OP == MAD64_PLUS_ADDC) {
          asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(c32[i]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == LSHL_ADD_U64) {
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
        } else if constexpr (OP == PK_MAD_U16) {
          asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(c32[i]) : "v"(a), "v"(b));
        } else if constexpr (OP == AND_OR) {
          asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c32[i]) : "v"(a), "v"(b));
        } else if constexpr (OP == CNDMASK) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c32[i]) : "v"(a) : );
        } else if constexpr (OP == LSHRREV_B64) {
          asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[i]));
        } else if constexpr (OP == ALIGNBIT) {
          asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == AND_B32) {
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == ADD_U32) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == SUB_U32) {
          asm volatile("v_sub_u32 %0, %0, %1" : "+v"(c32[i]) : "v"(a));
        } else if constexpr (OP == MAD_I64_I32) {
          asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (OP == BFE_U32) {
          asm volatile("v_bfe_u32 %0, %0, 3, 26" : "+v"(c32[i]));
        } else if constexpr (OP == CNDMASK_SGPR) {
          asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(c32[i]) : "v"(a), "s"(smask));
        } else if constexpr (OP == MOV_B32) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(c32[i]) : "v"(a));
        } else if constexpr (OP == MAD_U64_NOVCC) {
          asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(sdump) : "v"(a), "v"(b));
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int i = 0; i < UNROLL; ++i) s += acc[i] + (uint64_t)dacc[i] + (uint64_t)facc[i] + c32[i];
  out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ (uint32_t)sdump;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, uint32_t* d_out, unsigned long long* d_cyc, int n_cu) {
  const int trips = 64;
  int blocks = n_cu * 4 * waves_per_simd;   // 64-thread blocks, one wave each
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 64>>>(d_out, d_cyc, 4);   // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 64>>>(d_out, d_cyc, trips);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(blocks);
  CHECK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(cyc.begin(), cyc.end());
  double med = (double)cyc[blocks / 2];
  double instr_per_wave = (double)trips * INNER * UNROLL * ((OP == ADD_CO_PAIR) ? 2 : 1);
  double total_wave_instr = instr_per_wave * blocks;
  double ginstr = total_wave_instr * 64 / (ms * 1e-3) / 1e9;   // lane-ops per second (G)
  // s_memtime ticks at 100 MHz on gfx9 (constant clock); report wall-based numbers as primary.
  printf("%-44s w/SIMD=%d  time=%8.3f ms  %9.1f Glane-op/s  wave-instr/us/SIMD=%8.2f  memtime_ticks/wave=%.0f\n",
         op_names[OP], waves_per_simd, ms, ginstr, total_wave_instr / (n_cu * 4) / (ms * 1e3), med);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int OP> void sweep(uint32_t* d_out, unsigned long long* d_cyc, int n_cu) {
  for (int w : {1, 2, 4, 8}) run<OP>(w, d_out, d_cyc, n_cu);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  int n_cu = p.multiProcessorCount;
  printf("device: %s  CUs=%d  clockRate=%d kHz  arch=%s\n", p.name, n_cu, p.clockRate, p.gcnArchName);
  uint32_t* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 4 * 8 * 64 * sizeof(uint32_t)));
  CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 4 * 8 * sizeof(unsigned long long)));
  sweep<FMA_F32>(d_out, d_cyc, n_cu);
  sweep<MAD_U64_U32>(d_out, d_cyc, n_cu);
  sweep<MAD_U64_U32_DEP>(d_out, d_cyc, n_cu);
  sweep<MAD64_PLUS_ADDC>(d_out, d_cyc, n_cu);
  sweep<MUL_LO_U32>(d_out, d_cyc, n_cu);
  sweep<MUL_HI_U32>(d_out, d_cyc, n_cu);
  sweep<MAD_U32_U24>(d_out, d_cyc, n_cu);
  sweep<MUL_HI_U32_U24>(d_out, d_cyc, n_cu);
  sweep<ADD_CO_PAIR>(d_out, d_cyc, n_cu);
  sweep<ADD3_U32>(d_out, d_cyc, n_cu);
  sweep<LSHL_ADD_U64>(d_out, d_cyc, n_cu);
  sweep<PK_MAD_U16>(d_out, d_cyc, n_cu);
  sweep<AND_OR>(d_out, d_cyc, n_cu);
  sweep<CNDMASK>(d_out, d_cyc, n_cu);
  sweep<FMA_F64>(d_out, d_cyc, n_cu);
  sweep<LSHRREV_B64>(d_out, d_cyc, n_cu);
  sweep<ALIGNBIT>(d_out, d_cyc, n_cu);
  sweep<AND_B32>(d_out, d_cyc, n_cu);
  sweep<ADD_U32>(d_out, d_cyc, n_cu);
  sweep<SUB_U32>(d_out, d_cyc, n_cu);
  sweep<MAD_I64_I32>(d_out, d_cyc, n_cu);
  sweep<BFE_U32>(d_out, d_cyc, n_cu);
  sweep<CNDMASK_SGPR>(d_out, d_cyc, n_cu);
  sweep<MOV_B32>(d_out, d_cyc, n_cu);
  sweep<MAD_U64_NOVCC>(d_out, d_cyc, n_cu);
  return 0;
}
