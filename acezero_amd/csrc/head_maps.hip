// head_maps.hip -- the head's wide layers on whole frames (acez_head_forward_maps / acez_head_forward with >= 32 768 rows per pass:
// Regressor.forward at registration time, register_mapping.py:201-242 -> ace_network.py:120-149) as ONE launch. Included by head_api.hip.
//
// Until round 6 these passes ran the eight 512 -> 512 layers as eight launches of the encoder's large-tile kernel (convgemm512_kernel:
// 266-342 us per layer and 307 200 rows, 0.19-0.23 of the MFMA peak, every layer's activations written to and read back from HBM).
// Here a workgroup keeps a 128-row activation tile in LDS for the whole chain (128 KiB = rows x 512 channels x 16 bit, updated IN PLACE:
// a layer's output lives in the accumulators until every wave has finished reading its input) and pulls the weights straight into
// registers in MFMA operand order:
//   * Wf = fragment-ordered copy of the layers' 16-bit weights ([layer][32-column block][16-wide K step][lane][8]: a wave-instruction
//     loads 1 KiB contiguous; wfrag_pack_kernel, re-run at the start of every pass: 4 MiB, a few microseconds);
//   * wave w owns output columns 64 w .. 64 w + 63 of all 128 rows (2 x 4 accumulator blocks of 32 x 32), reads its A fragments (weights)
//     from L2 four K steps ahead and its B fragments (activations) from the tile, 16-byte chunk index XOR row & 31: conflict free;
//   * every row's dot products run over K in the same order whatever tile, wave or pass the row falls into: a frame's scene coordinates do
//     not depend on what else is in the batch (tests/test_session_gpu.py registers the same frames in two folder compositions and compares
//     pose files). The price is 7 %: started at a WORKGROUP-DEPENDENT K step (-DACEZ_HM_ROT=1: rot = (blockIdx >> 3) & 31), the 32
//     workgroups of an XCD read 32 different weight lines at any moment instead of queueing on one L2 channel -- 3.35 -> 3.13 ms per 136
//     frames (prototype, fully unrolled loop: 1765 -> 1506 us; with ONE weight panel for all layers, i.e. all CUs on the same 512 KiB: 2182)
//     -- but fp32 accumulation order would then depend on the tile's workgroup. Not taken;
//   * residual adds (ace_network.py:126,133: res = res + relu(conv(x)), the activation a 16-bit tensor before the add) as a second,
//     coalesced pass over the tile: the block input is re-read from global memory (block 0: the features themselves; later blocks: the
//     tile is copied out to R[b] when it is produced) -- a residual kept in registers would be 64 VGPRs next to 128 accumulators.
// tools/chain128_proto.hip is the timing prototype (8 layers on 307 200 rows: 1277 us against 2430 for the eight launches).
#pragma once
#include "gemm_common.h"

namespace acez {

struct HeadMapsArgs {
  const uint16_t* In;       // [n][512] feature rows
  const uint16_t* Wf;       // fragment-ordered weights of the L wide layers
  const float* params;      // the flat fp32 parameter vector (bias of layer l at l * 262656 + 262144)
  uint16_t* R[8];           // R[b], b = 1 .. nb: [n][512] scratch of the residual stream after block b - 1 (unused entries null)
  uint16_t* Out;            // [n][512]: the output of the last wide layer (fc2), what loss_kernel's fc3 reads -- or null with `fc3` below
  int n, nb;
  // fc3 + de-homogenisation on the finished tile (ace_network.py:135-149): the arithmetic of loss_body's phases A / B (same per-lane FMA
  // chains, the same DPP wave sum, the same expressions: the same function of the activations as loss_kernel's); the
  // [n][512] activation map is then neither written nor read (315 MB each way per 64 frames) and the separate launch (117 us) is gone.
  int fc3;                  // 1: write scene coordinates to out_xyz instead of activations to Out
  const uint16_t* W3;       // [no][512] 16-bit
  const float* b3;          // [no]
  int no, use_homogeneous;
  float mean[3], max_inv_scale, min_inv_scale, h_beta;
  float* out_xyz;           // [n][3], or with planar_hw > 0 [frames][3][planar_hw] (row m = frame * planar_hw + pixel), rows offset by row_offset
  int planar_hw, row_offset;
};

// Wf[((l * 16 + cb) * 32 + kk) * 512 + lane * 8 + e] = Wb[l][cb * 32 + (lane & 31)][kk * 16 + (lane >> 5) * 8 + e]
__global__ __launch_bounds__(256) void wfrag_pack_kernel(const uint16_t* __restrict__ Wb, uint16_t* __restrict__ Wf, int n_frags) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // fragment (l, cb, kk)
  if (f >= n_frags) return;
  const int kk = f & 31, cb = (f >> 5) & 15, l = f >> 9;
  const uint4 v = *reinterpret_cast<const uint4*>(Wb + (size_t)l * 262144 + (size_t)(cb * 32 + (lane & 31)) * 512 + kk * 16 + (lane >> 5) * 8);
  *reinterpret_cast<uint4*>(Wf + (size_t)f * 512 + lane * 8) = v;
}

template <class E>
__global__ __launch_bounds__(512) void head_maps_kernel(HeadMapsArgs a) {
  typedef typename E::frag frag;
  typedef __attribute__((address_space(3))) const frag lds_frag;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  __shared__ __attribute__((aligned(16))) uint16_t tile[128 * 512];
  __shared__ float s_s[128][4];               // fc3 outputs of the tile's rows
  lds_byte* const lds = (lds_byte*)tile;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = l & 31, fh = l >> 5;
  const int n = a.n, ntiles = (n + 127) >> 7;
  const int L = 3 * (a.nb + 1) + 2;
#ifndef ACEZ_HM_ROT
#define ACEZ_HM_ROT 0   // 1: timing variant (tools/lib_variant.sh), see the header comment
#endif
  const int rot = ACEZ_HM_ROT ? (int)((blockIdx.x >> 3) & 31) : 0;
  constexpr int PF = 4;
  unsigned baddr[4], bx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned r = j * 32 + fr;
    baddr[j] = r * 1024;
    bx[j] = (r & 31) ^ (unsigned)fh;
  }
  // the coalesced passes over the tile (copy out, residual add): thread t handles 16-byte chunks q = t + 512 i, row q >> 6, chunk q & 63
  auto copy_out = [&](uint16_t* dst, int m0) {
#pragma unroll 4
    for (int q = t; q < 128 * 64; q += 512) {
      const int row = q >> 6, ch = q & 63, m = m0 + row;
      if (m < n) *reinterpret_cast<uint4*>(dst + (size_t)m * 512 + ch * 8) = *reinterpret_cast<const uint4*>(&tile[row * 512 + ((ch ^ (row & 31)) << 3)]);
    }
  };
  // (Starting the workgroups of an XCD one K step apart in TIME -- same arithmetic order everywhere -- does not buy what the rotation buys:
  // 3.38 against 3.36 ms; the queue on the hot line re-synchronises them.)
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int m0 = tl << 7;
    // ---- input rows -> LDS by LDS-DMA: wave w rows 16 w .. 16 w + 15, one row (1 KiB) per instruction, lane = 16-byte chunk (swizzled source)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();               // the previous tile's copy-out has read the tile
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = w * 16 + i;
      const int g = min(m0 + row, n - 1);
      __builtin_amdgcn_global_load_lds((gvoid_t*)(a.In + (size_t)g * 512 + ((l ^ (row & 31)) << 3)), (lvoid_t*)(lds + row * 1024), 16, 0, 0);
    }
    ACEZ_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    for (int layer = 0; layer < L; ++layer) {
      f32x16 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      const uint16_t* wp[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wp[i] = a.Wf + ((size_t)(layer * 16 + w * 2 + i) * 32) * 512 + l * 8;
      frag wa[PF][2];
#pragma unroll
      for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) wa[k][i] = *reinterpret_cast<const frag*>(wp[i] + 512 * ((k + rot) & 31));
      frag fb[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[0][j] = *(lds_frag*)(lds + baddr[j] + ((((unsigned)(2 * rot)) ^ bx[j]) << 4));
      // K loop in groups of PF steps; the loop is not unrolled beyond a group and step addresses are computed as they are needed (fully
      // unrolled, the 32 steps kept ~190 address registers alive: 94 dwords of spills, 1506 us instead of 1277 in the prototype)
#pragma clang loop unroll(disable)
      for (int k0 = 0; k0 < 32; k0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int kk = k0 + u;
          const unsigned c2 = 2u * (unsigned)((kk + 1 + rot) & 31);       // (the read past the last step wraps to the first: unused)
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[(u + 1) & 1][j] = *(lds_frag*)(lds + baddr[j] + ((c2 ^ bx[j]) << 4));
          frag cur[2] = {wa[u][0], wa[u][1]};
          const int kn = (kk + PF + rot) & 31;                            // (past the end: early steps again, unused)
#pragma unroll
          for (int i = 0; i < 2; ++i) wa[u][i] = *reinterpret_cast<const frag*>(wp[i] + 512 * kn);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = E::mfma32(cur[i], fb[u & 1][j], acc[i][j]);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();             // every wave has read the layer's input: the tile may be overwritten
      {
        const float* bp = a.params + (int64_t)layer * 262656 + 262144 + w * 64 + 4 * fh;
        float4 bv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>(bp + i * 32 + 8 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ml = j * 32 + fr;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int nl = w * 64 + i * 32 + 8 * q + 4 * fh;
              const float4 b = bv[i][q];
              *reinterpret_cast<uint2*>(&tile[ml * 512 + ((((nl >> 3) ^ (ml & 31)) << 3) | (nl & 7))]) =
                  E::pk4(fmaxf(acc[i][j][4 * q + 0] + b.x, 0.f), fmaxf(acc[i][j][4 * q + 1] + b.y, 0.f), fmaxf(acc[i][j][4 * q + 2] + b.z, 0.f),
                         fmaxf(acc[i][j][4 * q + 3] + b.w, 0.f));
            }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();             // the layer's output tile is complete
      const int blk = layer / 3;
      if (layer < 3 * (a.nb + 1) && layer - 3 * blk == 2) {
        // ---- res = res + x (ace_network.py:126,133): tile := round16(tile + block input), the block input from global memory
        const uint16_t* rsrc = blk == 0 ? a.In : a.R[blk];
        uint4 rv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int q = t + 512 * i, row = q >> 6, ch = q & 63;
          rv[i] = *reinterpret_cast<const uint4*>(rsrc + (size_t)min(m0 + row, n - 1) * 512 + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int q = t + 512 * i, row = q >> 6, ch = q & 63;
          uint4* p = reinterpret_cast<uint4*>(&tile[row * 512 + ((ch ^ (row & 31)) << 3)]);
          const uint4 y = *p;
          float yf[8], rf[8];
          E::un4(make_uint2(y.x, y.y), yf); E::un4(make_uint2(y.z, y.w), yf + 4);
          E::un4(make_uint2(rv[i].x, rv[i].y), rf); E::un4(make_uint2(rv[i].z, rv[i].w), rf + 4);
          const uint2 lo = E::pk4(yf[0] + rf[0], yf[1] + rf[1], yf[2] + rf[2], yf[3] + rf[3]);
          const uint2 hi = E::pk4(yf[4] + rf[4], yf[5] + rf[5], yf[6] + rf[6], yf[7] + rf[7]);
          *p = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        if (blk < a.nb) copy_out(a.R[blk + 1], m0);   // the next block's residual input (each thread re-reads the chunks it wrote: no barrier)
      }
    }
    if (!a.fc3) {
      copy_out(a.Out, m0);
      continue;
    }
    // ---- fc3: wave w takes rows 16 w .. 16 w + 15, a row at a time: lane l holds channels 8 l .. 8 l + 7 (loss_body phase A)
    {
      const int no = a.no;
      float w3[4][8], b3v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 wr = *reinterpret_cast<const uint4*>(a.W3 + (size_t)(j < no ? j : 0) * 512 + l * 8);
        E::un4(make_uint2(wr.x, wr.y), &w3[j][0]);
        E::un4(make_uint2(wr.z, wr.w), &w3[j][4]);
        if (j >= no) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w3[j][e] = 0.f;
        }
        b3v[j] = (j < no) ? a.b3[j] : 0.f;
      }
#pragma unroll 4
      for (int rr = 0; rr < 16; ++rr) {
        const int r = w * 16 + rr;
        const uint4 xr = *reinterpret_cast<const uint4*>(&tile[r * 512 + ((l ^ (r & 31)) << 3)]);
        float x[8];
        E::un4(make_uint2(xr.x, xr.y), &x[0]);
        E::un4(make_uint2(xr.z, xr.w), &x[4]);
        float p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float sacc = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) sacc = fmaf(x[e], w3[j][e], sacc);
          p[j] = wave_sum63(sacc);
        }
        if (l == 63) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s_s[r][j] = (j < no) ? p[j] + b3v[j] : 0.f;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    // ---- de-homogenisation + mean (loss_body phase B's expressions), one thread per row, coalesced planar stores
    if (t < 128 && m0 + t < n) {
      const float s0 = s_s[t][0], s1 = s_s[t][1], s2 = s_s[t][2], s3 = s_s[t][3];
      float X[3];
      if (a.use_homogeneous) {
        const float bx = a.h_beta * s3;
        const float sp = (bx > 20.f) ? s3 : log1pf(expf(bx)) / a.h_beta;  // F.softplus(beta), ace_network.py:142
        const float hraw = sp + a.max_inv_scale;
        const float hval = (hraw > a.min_inv_scale) ? a.min_inv_scale : hraw;   // clamp_(max=min_inv_scale) :143
        X[0] = s0 / hval + a.mean[0];
        X[1] = s1 / hval + a.mean[1];
        X[2] = s2 / hval + a.mean[2];
      } else {
        X[0] = s0 + a.mean[0];
        X[1] = s1 + a.mean[1];
        X[2] = s2 + a.mean[2];
      }
      const int m = m0 + t;
      if (a.planar_hw > 0) {   // Regressor.forward's [B,3,H,W] layout (ace_network.py:265-270), what RANSAC reads
        const int64_t mg = (int64_t)a.row_offset + m;
        const int64_t fr2 = mg / a.planar_hw, px = mg - fr2 * a.planar_hw;
        float* o = a.out_xyz + fr2 * 3 * a.planar_hw + px;
        o[0] = X[0];
        o[(size_t)a.planar_hw] = X[1];
        o[(size_t)2 * a.planar_hw] = X[2];
      } else {
        a.out_xyz[(size_t)m * 3 + 0] = X[0];
        a.out_xyz[(size_t)m * 3 + 1] = X[1];
        a.out_xyz[(size_t)m * 3 + 2] = X[2];
      }
    }
  }
}

}  // namespace acez
