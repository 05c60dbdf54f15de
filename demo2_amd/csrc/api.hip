// C-ABI plumbing: error text, version, and the demo_gemm dispatcher.
#include "common.h"

thread_local char g_demo_err[512] = {0};

int demo_gemm_generic(const demo_gemm_desc& d, hipStream_t s);
int demo_gemm_bf16_eligible(const demo_gemm_desc& d);
int demo_gemm_bf16(const demo_gemm_desc& d, hipStream_t s);

extern "C" int demo_abi_version(void) { return 1; }
extern "C" const char* demo_last_error(void) { return g_demo_err; }
int demo_gemm_bf16_config(const demo_gemm_desc& d);
extern "C" int demo_gemm_path(const demo_gemm_desc* d) {
  if (!d || !demo_gemm_bf16_eligible(*d)) return 0;
  return demo_gemm_bf16_config(*d) == 2566 ? 2 : 1;
}
int demo_gemm_bf16_split_rows(const demo_gemm_desc& d);
extern "C" int demo_gemm_split_rows(const demo_gemm_desc* d) {
  if (!d || !demo_gemm_bf16_eligible(*d)) return 0;
  return demo_gemm_bf16_split_rows(*d);
}
int demo_gemm_bf16_split_plan(const demo_gemm_desc& d, int32_t* rows, int32_t* cfg, int max_parts);
extern "C" int demo_gemm_split_plan(const demo_gemm_desc* d, int32_t* rows, int32_t* cfg, int32_t max_parts) {
  if (!d || !rows || !cfg || max_parts < 1) return 0;
  if (!demo_gemm_bf16_eligible(*d)) { rows[0] = d->M; cfg[0] = 0; return 1; }
  return demo_gemm_bf16_split_plan(*d, rows, cfg, max_parts);
}
extern "C" int demo_gemm_desc_size(void) { return (int)sizeof(demo_gemm_desc); }

extern "C" int demo_gemm(const demo_gemm_desc* dp, void* stream) {
  DEMO_REQUIRE(dp != nullptr, "demo_gemm: null descriptor");
  demo_gemm_desc d = *dp;
  DEMO_REQUIRE(d.A && d.B && d.C, "demo_gemm: null operand");
  DEMO_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "demo_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
  DEMO_REQUIRE(d.batch1 >= 1 && d.batch2 >= 1 && (int64_t)d.batch1 * d.batch2 <= 65535, "demo_gemm: bad batch");
  DEMO_REQUIRE((d.dtA == DEMO_F32 || d.dtA == DEMO_BF16) && (d.dtB == DEMO_F32 || d.dtB == DEMO_BF16) &&
                   (d.dtC == DEMO_F32 || d.dtC == DEMO_BF16), "demo_gemm: bad dtype");
  DEMO_REQUIRE(!(d.accumulate && d.dtC != DEMO_F32), "demo_gemm: accumulate needs fp32 C");
  DEMO_REQUIRE(!d.pre_out || (d.pre_out_kind >= 0 && d.pre_out_kind <= 2), "demo_gemm: pre_out_kind %d (0, 1 or 2)", d.pre_out_kind);
  DEMO_REQUIRE(!d.act_grad_pre || d.dtPre == DEMO_F32 || d.dtPre == DEMO_BF16 || (d.dtPre == DEMO_D8 && d.act_grad_kind == DEMO_ACT_SAVED),
               "demo_gemm: dtPre %d with act_grad_kind %d (DEMO_D8 holds a saved derivative: DEMO_ACT_SAVED only)", d.dtPre, d.act_grad_kind);  hipStream_t s = (hipStream_t)stream;
  if (demo_gemm_bf16_eligible(d)) return demo_gemm_bf16(d, s);
  return demo_gemm_generic(d, s);
}
