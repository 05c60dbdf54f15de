// Probe (design study for round 3, NOT part of the library): the main loop of a 256 x 256 bf16 NT tile worked by FOUR waves of
// 128 x 128 (one wave per SIMD, 256 accumulator registers each) instead of the library's eight waves of 128 x 64 (two per SIMD,
// "ping-pong" halves).  What it is meant to answer (feed ablation): the 8-wave form reads 192 KB of LDS fragments
// per 64-deep K-tile and needs a barrier per 32 MFMAs; four square wave tiles read 128 KB, hide the LDS latency inside the wave
// (fragments of stage s+1 are requested under the MFMAs of stage s: two fragment sets, 128 registers) and need ONE barrier per stage.
//   LDS: ring of 5 stages x (A 256 x 32 + B 256 x 32 bf16 = 32 KB), LDS-DMA staging, two stages in flight beyond the one being read
//   (rows of 64 B, 16-B chunk index ^ S[(row >> 2) & 3], S = {0, 2, 3, 1}: conflict-free ds_read_b128, as in the library's twin kernel).
// The probe times C = A B^T on random data for the training step's shapes with a trivial epilogue (fp32 checksum of the accumulators
// per workgroup: the epilogue is NOT what is studied) and checks one tile against a host product.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pp4_probe.hip -o tools/probes/pp4_probe && tools/probes/pp4_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <cstring>
#include <cmath>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int BM = 256, BN = 256, BK = 32, NS = 5, OP_BYTES = 256 * BK * 2, STAGE = 2 * OP_BYTES;

__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }
__device__ __forceinline__ unsigned lds_off(const char* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// 256 rows x 64 B: 16 pieces of 1 KB (16 rows), 4 per wave
__device__ __forceinline__ void stage_op(const bf16_t* __restrict__ G, int64_t ld, int row0, int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = j * 4 + wave;
    const int r = piece * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz(r);
    const bf16_t* src = G + (int64_t)(row0 + r) * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + piece * 1024), 16, 0, 0);
  }
}
// eight ds_read_b128 of the fragment tiles 0..7 of one operand (1 KB apart), not waited for
__device__ __forceinline__ void read8(bf16x8 (&f)[8], unsigned base) {
  asm volatile("ds_read_b128 %0, %8\n\t"
               "ds_read_b128 %1, %8 offset:1024\n\t"
               "ds_read_b128 %2, %8 offset:2048\n\t"
               "ds_read_b128 %3, %8 offset:3072\n\t"
               "ds_read_b128 %4, %8 offset:4096\n\t"
               "ds_read_b128 %5, %8 offset:5120\n\t"
               "ds_read_b128 %6, %8 offset:6144\n\t"
               "ds_read_b128 %7, %8 offset:7168"
               : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
               : "v"(base)
               : "memory");
}

// ---- accumulators are OWNED BY INLINE ASM: a[0:255], tile (i, j) at a[32 i + 4 j : +3].  hipcc, left to itself, shuttles a 256-register
// accumulator set between the VGPR and AGPR halves (1451 v_accvgpr moves and 640 B of scratch in the first form of this probe).  The
// compiler never sees these registers as live: the kernel must stay below 256 VGPRs so that it has no reason to spill into them
// (audit: no v_accvgpr_* outside ASMSTART / ASMEND, scratch 0).

template <int C> __device__ __forceinline__ void mfma_chunk(const bf16x8 (&b)[8], const bf16x8& a) {
  if constexpr (C == 0)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], %0, %8, a[0:3]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[4:7], %1, %8, a[4:7]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[8:11], %2, %8, a[8:11]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[12:15], %3, %8, a[12:15]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[16:19], %4, %8, a[16:19]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[20:23], %5, %8, a[20:23]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[24:27], %6, %8, a[24:27]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[28:31], %7, %8, a[28:31]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
  else if constexpr (C == 1)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[32:35], %0, %8, a[32:35]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[36:39], %1, %8, a[36:39]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[40:43], %2, %8, a[40:43]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[44:47], %3, %8, a[44:47]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[48:51], %4, %8, a[48:51]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[52:55], %5, %8, a[52:55]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[56:59], %6, %8, a[56:59]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[60:63], %7, %8, a[60:63]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
  else if constexpr (C == 2)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[64:67], %0, %8, a[64:67]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[68:71], %1, %8, a[68:71]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[72:75], %2, %8, a[72:75]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[76:79], %3, %8, a[76:79]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[80:83], %4, %8, a[80:83]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[84:87], %5, %8, a[84:87]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[88:91], %6, %8, a[88:91]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[92:95], %7, %8, a[92:95]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
  else if constexpr (C == 3)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[96:99], %0, %8, a[96:99]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[100:103], %1, %8, a[100:103]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[104:107], %2, %8, a[104:107]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[108:111], %3, %8, a[108:111]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[112:115], %4, %8, a[112:115]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[116:119], %5, %8, a[116:119]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[120:123], %6, %8, a[120:123]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[124:127], %7, %8, a[124:127]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
  else if constexpr (C == 4)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[128:131], %0, %8, a[128:131]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[132:135], %1, %8, a[132:135]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[136:139], %2, %8, a[136:139]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[140:143], %3, %8, a[140:143]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[144:147], %4, %8, a[144:147]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[148:151], %5, %8, a[148:151]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[152:155], %6, %8, a[152:155]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[156:159], %7, %8, a[156:159]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
  else if constexpr (C == 5)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[160:163], %0, %8, a[160:163]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[164:167], %1, %8, a[164:167]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[168:171], %2, %8, a[168:171]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[172:175], %3, %8, a[172:175]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[176:179], %4, %8, a[176:179]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[180:183], %5, %8, a[180:183]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[184:187], %6, %8, a[184:187]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[188:191], %7, %8, a[188:191]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
  else if constexpr (C == 6)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[192:195], %0, %8, a[192:195]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[196:199], %1, %8, a[196:199]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[200:203], %2, %8, a[200:203]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[204:207], %3, %8, a[204:207]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[208:211], %4, %8, a[208:211]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[212:215], %5, %8, a[212:215]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[216:219], %6, %8, a[216:219]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[220:223], %7, %8, a[220:223]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
  else if constexpr (C == 7)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[224:227], %0, %8, a[224:227]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[228:231], %1, %8, a[228:231]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[232:235], %2, %8, a[232:235]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[236:239], %3, %8, a[236:239]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[240:243], %4, %8, a[240:243]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[244:247], %5, %8, a[244:247]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[248:251], %6, %8, a[248:251]\n\t"
                 "v_mfma_f32_16x16x32_bf16 a[252:255], %7, %8, a[252:255]\n\t"
                 :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(a)
                 : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\t" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
  asm volatile("v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\t" ::: "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
  asm volatile("v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\t" ::: "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
  asm volatile("v_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0\n\t" ::: "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
  asm volatile("v_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0\n\tv_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0\n\t" ::: "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
  asm volatile("v_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0\n\tv_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0\n\t" ::: "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
  asm volatile("v_accvgpr_write_b32 a192, 0\n\tv_accvgpr_write_b32 a193, 0\n\tv_accvgpr_write_b32 a194, 0\n\tv_accvgpr_write_b32 a195, 0\n\tv_accvgpr_write_b32 a196, 0\n\tv_accvgpr_write_b32 a197, 0\n\tv_accvgpr_write_b32 a198, 0\n\tv_accvgpr_write_b32 a199, 0\n\tv_accvgpr_write_b32 a200, 0\n\tv_accvgpr_write_b32 a201, 0\n\tv_accvgpr_write_b32 a202, 0\n\tv_accvgpr_write_b32 a203, 0\n\tv_accvgpr_write_b32 a204, 0\n\tv_accvgpr_write_b32 a205, 0\n\tv_accvgpr_write_b32 a206, 0\n\tv_accvgpr_write_b32 a207, 0\n\tv_accvgpr_write_b32 a208, 0\n\tv_accvgpr_write_b32 a209, 0\n\tv_accvgpr_write_b32 a210, 0\n\tv_accvgpr_write_b32 a211, 0\n\tv_accvgpr_write_b32 a212, 0\n\tv_accvgpr_write_b32 a213, 0\n\tv_accvgpr_write_b32 a214, 0\n\tv_accvgpr_write_b32 a215, 0\n\tv_accvgpr_write_b32 a216, 0\n\tv_accvgpr_write_b32 a217, 0\n\tv_accvgpr_write_b32 a218, 0\n\tv_accvgpr_write_b32 a219, 0\n\tv_accvgpr_write_b32 a220, 0\n\tv_accvgpr_write_b32 a221, 0\n\tv_accvgpr_write_b32 a222, 0\n\tv_accvgpr_write_b32 a223, 0\n\t" ::: "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
  asm volatile("v_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\tv_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\tv_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\tv_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0\n\tv_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\tv_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\tv_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\tv_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0\n\t" ::: "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}
template <int T> __device__ __forceinline__ f32x4 acc_read() {
  f32x4 v;
  if constexpr (T == 0) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 1) asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 2) asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 3) asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_accvgpr_read_b32 %2, a14\n\tv_accvgpr_read_b32 %3, a15" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 4) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 5) asm volatile("v_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a22\n\tv_accvgpr_read_b32 %3, a23" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 6) asm volatile("v_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a26\n\tv_accvgpr_read_b32 %3, a27" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 7) asm volatile("v_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a30\n\tv_accvgpr_read_b32 %3, a31" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 8) asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 9) asm volatile("v_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a39" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 10) asm volatile("v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a43" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 11) asm volatile("v_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a47" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 12) asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 13) asm volatile("v_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\tv_accvgpr_read_b32 %2, a54\n\tv_accvgpr_read_b32 %3, a55" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 14) asm volatile("v_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\tv_accvgpr_read_b32 %2, a58\n\tv_accvgpr_read_b32 %3, a59" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 15) asm volatile("v_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\tv_accvgpr_read_b32 %2, a62\n\tv_accvgpr_read_b32 %3, a63" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 16) asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 17) asm volatile("v_accvgpr_read_b32 %0, a68\n\tv_accvgpr_read_b32 %1, a69\n\tv_accvgpr_read_b32 %2, a70\n\tv_accvgpr_read_b32 %3, a71" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 18) asm volatile("v_accvgpr_read_b32 %0, a72\n\tv_accvgpr_read_b32 %1, a73\n\tv_accvgpr_read_b32 %2, a74\n\tv_accvgpr_read_b32 %3, a75" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 19) asm volatile("v_accvgpr_read_b32 %0, a76\n\tv_accvgpr_read_b32 %1, a77\n\tv_accvgpr_read_b32 %2, a78\n\tv_accvgpr_read_b32 %3, a79" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 20) asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 21) asm volatile("v_accvgpr_read_b32 %0, a84\n\tv_accvgpr_read_b32 %1, a85\n\tv_accvgpr_read_b32 %2, a86\n\tv_accvgpr_read_b32 %3, a87" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 22) asm volatile("v_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\tv_accvgpr_read_b32 %2, a90\n\tv_accvgpr_read_b32 %3, a91" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 23) asm volatile("v_accvgpr_read_b32 %0, a92\n\tv_accvgpr_read_b32 %1, a93\n\tv_accvgpr_read_b32 %2, a94\n\tv_accvgpr_read_b32 %3, a95" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 24) asm volatile("v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\tv_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 25) asm volatile("v_accvgpr_read_b32 %0, a100\n\tv_accvgpr_read_b32 %1, a101\n\tv_accvgpr_read_b32 %2, a102\n\tv_accvgpr_read_b32 %3, a103" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 26) asm volatile("v_accvgpr_read_b32 %0, a104\n\tv_accvgpr_read_b32 %1, a105\n\tv_accvgpr_read_b32 %2, a106\n\tv_accvgpr_read_b32 %3, a107" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 27) asm volatile("v_accvgpr_read_b32 %0, a108\n\tv_accvgpr_read_b32 %1, a109\n\tv_accvgpr_read_b32 %2, a110\n\tv_accvgpr_read_b32 %3, a111" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 28) asm volatile("v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\tv_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 29) asm volatile("v_accvgpr_read_b32 %0, a116\n\tv_accvgpr_read_b32 %1, a117\n\tv_accvgpr_read_b32 %2, a118\n\tv_accvgpr_read_b32 %3, a119" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 30) asm volatile("v_accvgpr_read_b32 %0, a120\n\tv_accvgpr_read_b32 %1, a121\n\tv_accvgpr_read_b32 %2, a122\n\tv_accvgpr_read_b32 %3, a123" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 31) asm volatile("v_accvgpr_read_b32 %0, a124\n\tv_accvgpr_read_b32 %1, a125\n\tv_accvgpr_read_b32 %2, a126\n\tv_accvgpr_read_b32 %3, a127" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 32) asm volatile("v_accvgpr_read_b32 %0, a128\n\tv_accvgpr_read_b32 %1, a129\n\tv_accvgpr_read_b32 %2, a130\n\tv_accvgpr_read_b32 %3, a131" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 33) asm volatile("v_accvgpr_read_b32 %0, a132\n\tv_accvgpr_read_b32 %1, a133\n\tv_accvgpr_read_b32 %2, a134\n\tv_accvgpr_read_b32 %3, a135" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 34) asm volatile("v_accvgpr_read_b32 %0, a136\n\tv_accvgpr_read_b32 %1, a137\n\tv_accvgpr_read_b32 %2, a138\n\tv_accvgpr_read_b32 %3, a139" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 35) asm volatile("v_accvgpr_read_b32 %0, a140\n\tv_accvgpr_read_b32 %1, a141\n\tv_accvgpr_read_b32 %2, a142\n\tv_accvgpr_read_b32 %3, a143" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 36) asm volatile("v_accvgpr_read_b32 %0, a144\n\tv_accvgpr_read_b32 %1, a145\n\tv_accvgpr_read_b32 %2, a146\n\tv_accvgpr_read_b32 %3, a147" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 37) asm volatile("v_accvgpr_read_b32 %0, a148\n\tv_accvgpr_read_b32 %1, a149\n\tv_accvgpr_read_b32 %2, a150\n\tv_accvgpr_read_b32 %3, a151" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 38) asm volatile("v_accvgpr_read_b32 %0, a152\n\tv_accvgpr_read_b32 %1, a153\n\tv_accvgpr_read_b32 %2, a154\n\tv_accvgpr_read_b32 %3, a155" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 39) asm volatile("v_accvgpr_read_b32 %0, a156\n\tv_accvgpr_read_b32 %1, a157\n\tv_accvgpr_read_b32 %2, a158\n\tv_accvgpr_read_b32 %3, a159" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 40) asm volatile("v_accvgpr_read_b32 %0, a160\n\tv_accvgpr_read_b32 %1, a161\n\tv_accvgpr_read_b32 %2, a162\n\tv_accvgpr_read_b32 %3, a163" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 41) asm volatile("v_accvgpr_read_b32 %0, a164\n\tv_accvgpr_read_b32 %1, a165\n\tv_accvgpr_read_b32 %2, a166\n\tv_accvgpr_read_b32 %3, a167" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 42) asm volatile("v_accvgpr_read_b32 %0, a168\n\tv_accvgpr_read_b32 %1, a169\n\tv_accvgpr_read_b32 %2, a170\n\tv_accvgpr_read_b32 %3, a171" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 43) asm volatile("v_accvgpr_read_b32 %0, a172\n\tv_accvgpr_read_b32 %1, a173\n\tv_accvgpr_read_b32 %2, a174\n\tv_accvgpr_read_b32 %3, a175" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 44) asm volatile("v_accvgpr_read_b32 %0, a176\n\tv_accvgpr_read_b32 %1, a177\n\tv_accvgpr_read_b32 %2, a178\n\tv_accvgpr_read_b32 %3, a179" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 45) asm volatile("v_accvgpr_read_b32 %0, a180\n\tv_accvgpr_read_b32 %1, a181\n\tv_accvgpr_read_b32 %2, a182\n\tv_accvgpr_read_b32 %3, a183" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 46) asm volatile("v_accvgpr_read_b32 %0, a184\n\tv_accvgpr_read_b32 %1, a185\n\tv_accvgpr_read_b32 %2, a186\n\tv_accvgpr_read_b32 %3, a187" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 47) asm volatile("v_accvgpr_read_b32 %0, a188\n\tv_accvgpr_read_b32 %1, a189\n\tv_accvgpr_read_b32 %2, a190\n\tv_accvgpr_read_b32 %3, a191" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 48) asm volatile("v_accvgpr_read_b32 %0, a192\n\tv_accvgpr_read_b32 %1, a193\n\tv_accvgpr_read_b32 %2, a194\n\tv_accvgpr_read_b32 %3, a195" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 49) asm volatile("v_accvgpr_read_b32 %0, a196\n\tv_accvgpr_read_b32 %1, a197\n\tv_accvgpr_read_b32 %2, a198\n\tv_accvgpr_read_b32 %3, a199" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 50) asm volatile("v_accvgpr_read_b32 %0, a200\n\tv_accvgpr_read_b32 %1, a201\n\tv_accvgpr_read_b32 %2, a202\n\tv_accvgpr_read_b32 %3, a203" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 51) asm volatile("v_accvgpr_read_b32 %0, a204\n\tv_accvgpr_read_b32 %1, a205\n\tv_accvgpr_read_b32 %2, a206\n\tv_accvgpr_read_b32 %3, a207" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 52) asm volatile("v_accvgpr_read_b32 %0, a208\n\tv_accvgpr_read_b32 %1, a209\n\tv_accvgpr_read_b32 %2, a210\n\tv_accvgpr_read_b32 %3, a211" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 53) asm volatile("v_accvgpr_read_b32 %0, a212\n\tv_accvgpr_read_b32 %1, a213\n\tv_accvgpr_read_b32 %2, a214\n\tv_accvgpr_read_b32 %3, a215" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 54) asm volatile("v_accvgpr_read_b32 %0, a216\n\tv_accvgpr_read_b32 %1, a217\n\tv_accvgpr_read_b32 %2, a218\n\tv_accvgpr_read_b32 %3, a219" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 55) asm volatile("v_accvgpr_read_b32 %0, a220\n\tv_accvgpr_read_b32 %1, a221\n\tv_accvgpr_read_b32 %2, a222\n\tv_accvgpr_read_b32 %3, a223" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 56) asm volatile("v_accvgpr_read_b32 %0, a224\n\tv_accvgpr_read_b32 %1, a225\n\tv_accvgpr_read_b32 %2, a226\n\tv_accvgpr_read_b32 %3, a227" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 57) asm volatile("v_accvgpr_read_b32 %0, a228\n\tv_accvgpr_read_b32 %1, a229\n\tv_accvgpr_read_b32 %2, a230\n\tv_accvgpr_read_b32 %3, a231" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 58) asm volatile("v_accvgpr_read_b32 %0, a232\n\tv_accvgpr_read_b32 %1, a233\n\tv_accvgpr_read_b32 %2, a234\n\tv_accvgpr_read_b32 %3, a235" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 59) asm volatile("v_accvgpr_read_b32 %0, a236\n\tv_accvgpr_read_b32 %1, a237\n\tv_accvgpr_read_b32 %2, a238\n\tv_accvgpr_read_b32 %3, a239" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 60) asm volatile("v_accvgpr_read_b32 %0, a240\n\tv_accvgpr_read_b32 %1, a241\n\tv_accvgpr_read_b32 %2, a242\n\tv_accvgpr_read_b32 %3, a243" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 61) asm volatile("v_accvgpr_read_b32 %0, a244\n\tv_accvgpr_read_b32 %1, a245\n\tv_accvgpr_read_b32 %2, a246\n\tv_accvgpr_read_b32 %3, a247" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 62) asm volatile("v_accvgpr_read_b32 %0, a248\n\tv_accvgpr_read_b32 %1, a249\n\tv_accvgpr_read_b32 %2, a250\n\tv_accvgpr_read_b32 %3, a251" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  else if constexpr (T == 63) asm volatile("v_accvgpr_read_b32 %0, a252\n\tv_accvgpr_read_b32 %1, a253\n\tv_accvgpr_read_b32 %2, a254\n\tv_accvgpr_read_b32 %3, a255" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
  return v;
}
template <int T = 0, typename F> __device__ __forceinline__ void acc_for_each(F&& f) {
  if constexpr (T < 64) {
    f(T >> 3, T & 7, acc_read<T>());
    acc_for_each<T + 1>(f);
  }
}

// MODE 0: full kernel; 1: no LDS-DMA inside the loop (what the loop costs without a feed)
template <int MODE>
__global__ __launch_bounds__(256, 1) void pp4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ C,
                                                     float* __restrict__ sums, int M, int N, int K, int ntn, int group_m) {
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order as in the library: groups of group_m row panels against all column panels
  const int ntm = M / BM, nwg = ntm * ntn, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  const int tid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int per_group = group_m * ntn;
  const int grp = tid / per_group, in_grp = tid - grp * per_group;
  const int gm0 = grp * group_m;
  const int gsz = (ntm - gm0) < group_m ? (ntm - gm0) : group_m;
  const int m0 = (gm0 + in_grp % gsz) * BM, n0 = (in_grp / gsz) * BN;

  const int frow = lane & 15, g4 = lane >> 4;
  const int foff = frow * 64 + ((g4 ^ swz(frow)) << 4);
  const int nk = K / BK;

  acc_zero();

  auto stage = [&](int s) {
    char* buf = smem + (s % NS) * STAGE;
    stage_op(A, K, m0, s * BK, buf, wave, lane);
    stage_op(B, K, n0, s * BK, buf + OP_BYTES, wave, lane);
  };
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][8], bfr[2][8];                     // two fragment sets: the one being multiplied, the one arriving
  {
    const unsigned base = lds_off(smem + foff);
    read8(af[0], base + wr * (128 * 64));
    read8(bfr[0], base + OP_BYTES + wc * (128 * 64));
  }
  int slot = 0;
  // one stage: 8 chunks of {one LDS-DMA piece of stage s+3, two fragment reads of stage s+1, 8 MFMAs of stage s}; with ONE wave per SIMD
  // nothing else can run while this wave issues its loads, so they are dealt out between the MFMAs (an MFMA holds the issue port for
  // 8 of its 16 clocks) instead of being issued in a block in front of them
  auto body = [&](int s, auto curc) {
    constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
    const bool more = s + 1 < nk, feed = MODE == 0 && s + 3 < nk;
    // stage s+1 must have landed (own pieces), then be visible to every wave, before its fragments are requested below; the same
    // barrier tells that every wave has finished reading stage s-2, whose slot (s+3) % 5 is re-staged below
    if (more) {
      if (MODE == 0) {
        if (s + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // stage s+2 (8 pieces) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // the fragments of stage s (requested one stage ago)
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    const int ns = slot + 1 == NS ? 0 : slot + 1;
    const unsigned nbase = lds_off(smem + ns * STAGE + foff);
    char* dbuf = smem + ((s + 3) % NS) * STAGE;
    auto chunk = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (feed) {                                   // piece c of stage s+3: A pieces 0..3, then B pieces 0..3
        const int j = c & 3;
        const int piece = j * 4 + wave;
        const int r = piece * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ swz(r);
        const bf16_t* src = (c < 4 ? A + (int64_t)(m0 + r) * K : B + (int64_t)(n0 + r) * K) + (s + 3) * BK + ch * 8;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dbuf + (c < 4 ? 0 : OP_BYTES) + piece * 1024), 16, 0, 0);
      }
      if (more) {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"
                     : "=&v"(af[nxt][c]), "=&v"(bfr[nxt][c])
                     : "v"(nbase + wr * (128 * 64) + c * 1024), "v"(nbase + OP_BYTES + wc * (128 * 64) + c * 1024)
                     : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk<c>(bfr[cur], af[cur][c]);
      __builtin_amdgcn_sched_barrier(0);
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{}); chunk(std::integral_constant<int, 3>{});
    chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{}); chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{});
    slot = ns;
  };
  for (int s = 0; s < nk; s += 2) {
    body(s, std::integral_constant<int, 0>{});
    if (s + 1 < nk) body(s + 1, std::integral_constant<int, 1>{});
  }
  // trivial epilogue: one tile optionally written (check), otherwise a checksum per lane (keeps the accumulators alive)
  // D = mfma(B-frag, A-frag): D[row = n (4 g4 + r)][col = m (frow)]
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");             // (MFMA results -> v_accvgpr_read: let the last MFMAs retire)
  float t = 0.f;
  const bool wr_c = C && bid == 0;
  acc_for_each([&](int i, int j, f32x4 v) {
    if (wr_c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 128 + i * 16 + frow, n = n0 + wc * 128 + j * 16 + 4 * g4 + r;
        C[(int64_t)m * N + n] = v[r];
      }
    }
    t += (v[0] + v[1]) + (v[2] + v[3]);
  });
  sums[(int64_t)bid * 256 + threadIdx.x] = t;
}

static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

template <int MODE> float run(const bf16_t* A, const bf16_t* B, float* C, float* sums, int M, int N, int K, int reps) {
  const int ntn = N / BN, ntm = M / BM;
  int gm = 32 / ntn; gm = gm < 1 ? 1 : (gm > 8 ? 8 : gm);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(pp4_kernel<MODE>, dim3(ntm * ntn), dim3(256), 0, 0, A, B, C, sums, M, N, K, ntn, gm);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(pp4_kernel<MODE>, dim3(ntm * ntn), dim3(256), 0, 0, A, B, (float*)nullptr, sums, M, N, K, ntn, gm);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const int M = 49408;                             // 193 row panels of 256 (the step's 49536 rows without the half panel)
  const int shapes[][2] = {{2304, 768}, {3072, 768}, {768, 3072}, {768, 2304}, {768, 768}, {3072, 1536}, {3072, 384}};
  const int maxN = 3072, maxK = 3072;
  std::vector<bf16_t> hA((size_t)M * maxK), hB((size_t)maxN * maxK);
  srand(1);
  for (auto& v : hA) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
  for (auto& v : hB) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
  bf16_t *A, *B; float *C, *sums;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&B, hB.size() * 2);
  hipMalloc(&C, (size_t)M * maxN * 4); hipMalloc(&sums, (size_t)4096 * 256 * 4);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  // correctness of one tile (N = 768, K = 768): workgroup 0's tile against a host product
  {
    const int N = 768, K = 768;
    hipMemset(C, 0, (size_t)M * N * 4);
    run<0>(A, B, C, sums, M, N, K, 1);
    std::vector<float> hC((size_t)256 * N);
    hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    // workgroup 0 -> tile (m0 = 0, n0 = 0) under this order
    double maxerr = 0.0, maxref = 0.0;
    for (int m = 0; m < 256; m += 37)
      for (int n = 0; n < 256; n += 29) {
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
        const double err = fabs(ref - hC[(size_t)m * N + n]);
        if (err > maxerr) maxerr = err;
        if (fabs(ref) > maxref) maxref = fabs(ref);
      }
    printf("check (N=768, K=768, tile 0): max |err| %.3e against max |ref| %.3e  %s\n", maxerr, maxref, maxerr <= 1e-3 * (maxref + 1.0) ? "OK" : "MISMATCH");
  }
  printf("%-28s %12s %12s\n", "M=49408  shape", "pp4 loop", "pp4, no feed");
  for (auto& sh : shapes) {
    const int N = sh[0], K = sh[1];
    // note: A and B are read with row stride K (a K-column window of the buffers): same bytes as a dense operand
    const float t0 = run<0>(A, B, C, sums, M, N, K, 10), t1 = run<1>(A, B, C, sums, M, N, K, 10);
    const double fl = 2.0 * M * N * K;
    printf("N=%4d K=%4d %14.1f us %6.0f TF %10.1f us %6.0f TF   (%.2f rounds of 256 CUs)\n", N, K, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6,
           (double)(M / 256) * (N / 256) / 256.0);
  }
  return 0;
}
