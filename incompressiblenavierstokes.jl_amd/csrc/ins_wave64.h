// Wave-level helpers shared by the 64-outputs-per-wavefront stage kernels (ins_flux64.hip, ins_flux64m.hip): lane reads, DPP wave shifts with a
// halo value for the lane that has no source lane, and buffer-resource addressing (descriptor = one plane of one array, soffset = row start
// held in an SGPR, voffset = the lane's column: all plane / row arithmetic runs on the scalar unit).
#pragma once
#include "ins_internal.h"

namespace {

__device__ __forceinline__ double rdlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <int CTRL>
__device__ __forceinline__ double dpp_old(double old, double v) {  // lanes without a source lane keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_old(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <typename T>
__device__ __forceinline__ T next_h(T v, T h) { return dpp_old<0x130>(h, v); }  // lane l <- l+1, lane 63 <- h
template <typename T>
__device__ __forceinline__ T prev_h(T v, T h) { return dpp_old<0x138>(h, v); }  // lane l <- l-1, lane 0  <- h

// Buffer addressing: descriptor (4 SGPRs) = one plane of one array, soffset (SGPR) = row start, voffset (VGPR) = column.
// All plane / row arithmetic runs on the scalar unit; a lane holds one 32-bit offset for every load and store it issues.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
template <typename T>
__device__ __forceinline__ rsrc_t plane_rsrc(const T* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ T ldb(rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ double ldb<double>(rsrc_t r, unsigned voff, unsigned soff) {
  const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
template <>
__device__ __forceinline__ float ldb<float>(rsrc_t r, unsigned voff, unsigned soff) {
  return __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// loads with the cache-policy bits of the instruction (aux 2 = nt): arrays a kernel reads exactly once (the epilogue terms of the stage kernels) — an experiment switch
template <int AUX>
__device__ __forceinline__ double ldb_aux(rsrc_t r, unsigned voff, unsigned soff, double) {
  const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
  return __hiloint2double((int)v.y, (int)v.x);
}
template <int AUX>
__device__ __forceinline__ float ldb_aux(rsrc_t r, unsigned voff, unsigned soff, float) {
  return __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
__device__ __forceinline__ void stb(rsrc_t r, unsigned voff, unsigned soff, double x) {
  v2u v;
  v.x = (unsigned)__double2loint(x);
  v.y = (unsigned)__double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// the same with the cache-policy bits of the instruction (aux: 1 = sc0, 2 = nt, 16 = sc1): an experiment switch (INS_FLUX64_NT)
template <int AUX>
__device__ __forceinline__ void stb_aux(rsrc_t r, unsigned voff, unsigned soff, double x) {
  v2u v;
  v.x = (unsigned)__double2loint(x);
  v.y = (unsigned)__double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void stb_aux(rsrc_t r, unsigned voff, unsigned soff, float x) {
  __builtin_amdgcn_raw_buffer_store_b32((unsigned)__float_as_int(x), r, voff, soff, AUX);
}
__device__ __forceinline__ void stb(rsrc_t r, unsigned voff, unsigned soff, float x) {
  __builtin_amdgcn_raw_buffer_store_b32((unsigned)__float_as_int(x), r, voff, soff, 0);
}


__device__ __forceinline__ int wrapi(int q, int n) {  // q in [-n, 2n) -> [0, n)
  return q < 0 ? q + n : (q >= n ? q - n : q);
}

}  // namespace
