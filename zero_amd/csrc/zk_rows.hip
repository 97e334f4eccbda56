// zk_rows.hip -- row-sparse exchange of an embedding-table gradient between data-parallel ranks.
//
// The reference's towers hand the source-embedding gradient around as tf.IndexedSlices: values and indices are
// concatenated across the towers and de-duplicated (utils/parallel.py:142-181) -- only the rows a batch touched
// travel.  Here every rank packs the rows ITS batch touched (<= tokens of the batch, out of 32000) into a send
// buffer, the ranks all-gather the packed buffers, and every rank adds the N payloads into its (zeroed) dense table in
// rank order.  All ranks therefore compute bit-identical tables; with fp32 payloads the result equals the dense sum
// all-reduce up to fp32 summation order.
//
// Payload of one rank: [ids int32 x R][rows x R x H] with R a capacity agreed at start-up; unused slots carry id -1;
// a batch with more than R distinct ids poisons its payload with NaN (k_rows_pack) -- a loud, collective failure.
// HBM-bound, R x H elements per launch (4096 x 512: 8 MB read, 4-8 MB written).
#include "zk_common.h"

extern "C" {
int zk_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, hipStream_t stream);
int zk_rows_pack(float* dtable, const int* uid, const int* n_uniq_dev, void* out, int R, int H, int out_bf16,
                 int clear_rows, hipStream_t stream);
int zk_rows_scatter_add(float* dtable, const void* payload, int R, int H, int in_bf16, int vocab_rows,
                        hipStream_t stream);
size_t zk_rows_payload_bytes(int R, int H, int bf16);
}

// one wave per packed slot; slot u < n: ids[u] = uid[u], rows[u] = cast(dtable[uid[u]]) (and the table row is cleared
// so that the scatter pass rebuilds the table from the payloads of ALL ranks, this one's included: every rank then adds
// the same rounded values in the same order); slots >= n: id -1, zero row
template <bool BF16>
__global__ void __launch_bounds__(256) k_rows_pack(float* __restrict__ dtable, const int* __restrict__ uid,
                                                   const int* __restrict__ n_uniq_dev, int* __restrict__ ids,
                                                   void* __restrict__ rows, int R, int H, int clear_rows) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int n_all = *n_uniq_dev;
  const int nu = min(n_all, R);
  // More distinct ids than slots: the rows beyond R cannot travel.  No rank can tell the others in time (they are
  // already in, or about to enter, the all-gather), so the failure is made loud ON THE DEVICE: slot 0 carries NaN, every
  // rank adds it into its table, the gradient norm is NaN on every rank and the loop stops at its NaN check
  // (main.py:316-319) instead of training on a silently truncated gradient -- or hanging on a rank-local exception.
  const bool overflow = n_all > R;
  for (int u = wave; u < R; u += nwaves) {
    const bool live = u < nu;
    const int id = live ? uid[u] : -1;
    if (lane == 0) ids[u] = id;
    float* src = dtable + (size_t)(live ? id : 0) * H;
    for (int c = lane * 4; c < H; c += 64 * 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        v = *reinterpret_cast<const float4*>(src + c);
        if (clear_rows) *reinterpret_cast<float4*>(src + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (overflow && u == 0) v.x = __uint_as_float(0x7fc00000u);
      }
      if (BF16)
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(rows) + (size_t)u * H + c) =
            make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
      else
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(rows) + (size_t)u * H + c) = v;
    }
  }
}

// dtable[ids[u]] += rows[u] for every slot with a valid id.  Ids are unique inside one payload, so the
// read-modify-write needs no atomics; payloads of different ranks are added by consecutive launches (fixed order).
template <bool BF16>
__global__ void __launch_bounds__(256) k_rows_scatter_add(float* __restrict__ dtable, const int* __restrict__ ids,
                                                          const void* __restrict__ rows, int R, int H,
                                                          int vocab_rows) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int u = wave; u < R; u += nwaves) {
    const int id = ids[u];
    if (id < 0 || id >= vocab_rows) continue;
    float* dst = dtable + (size_t)id * H;
    for (int c = lane * 4; c < H; c += 64 * 4) {
      float4 v;
      if (BF16) {
        const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(rows) + (size_t)u * H + c);
        v = make_float4(__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16),
                        __uint_as_float(p.y & 0xffff0000u));
      } else {
        v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rows) + (size_t)u * H + c);
      }
      float4 a = *reinterpret_cast<float4*>(dst + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      *reinterpret_cast<float4*>(dst + c) = a;
    }
  }
}

// dst[c][r] = src[r][c] on bf16 matrices through a 64 x 64 LDS tile (+1 column of padding: conflict-free both ways).
// Used once per weight version for the operand layout of the fused decode kernels (a projection weight with the input
// dimension contiguous); HBM-bound, 4 bytes per element.
__global__ void __launch_bounds__(256) k_transpose_bf16(const bf16_t* __restrict__ src, int ld_src, bf16_t* __restrict__ dst,
                                                        int ld_dst, int rows, int cols) {
  __shared__ bf16_t tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(size_t)c * ld_dst + r] = tile[tx][i];
  }
}

extern "C" {

int zk_transpose_bf16(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, hipStream_t stream) {
  ZK_CHECK_ARG(src != nullptr && dst != nullptr && ld_src >= cols && ld_dst >= rows, "zk_transpose_bf16: bad arguments");
  if (rows == 0 || cols == 0) return 0;
  hipLaunchKernelGGL(k_transpose_bf16, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, stream,
                     (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols);
  ZK_LAUNCH_CHECK();
  return 0;
}

size_t zk_rows_payload_bytes(int R, int H, int bf16) {
  return (size_t)R * 4 + (size_t)R * H * (bf16 ? 2 : 4);
}

int zk_rows_pack(float* dtable, const int* uid, const int* n_uniq_dev, void* out, int R, int H, int out_bf16,
                 int clear_rows, hipStream_t stream) {
  ZK_CHECK_ARG(H % 4 == 0 && R % 4 == 0, "zk_rows_pack: H=%d and R=%d must be multiples of 4", H, R);
  ZK_CHECK_ARG(dtable != nullptr && uid != nullptr && n_uniq_dev != nullptr && out != nullptr, "zk_rows_pack: null pointer");
  if (R == 0) return 0;
  int g = (R + 3) / 4;
  if (g > 2048) g = 2048;
  int* ids = reinterpret_cast<int*>(out);
  void* rows = reinterpret_cast<unsigned char*>(out) + (size_t)R * 4;
  if (out_bf16)
    hipLaunchKernelGGL(k_rows_pack<true>, dim3(g), dim3(256), 0, stream, dtable, uid, n_uniq_dev, ids, rows, R, H, clear_rows);
  else
    hipLaunchKernelGGL(k_rows_pack<false>, dim3(g), dim3(256), 0, stream, dtable, uid, n_uniq_dev, ids, rows, R, H, clear_rows);
  ZK_LAUNCH_CHECK();
  return 0;
}

int zk_rows_scatter_add(float* dtable, const void* payload, int R, int H, int in_bf16, int vocab_rows,
                        hipStream_t stream) {
  ZK_CHECK_ARG(H % 4 == 0 && R % 4 == 0, "zk_rows_scatter_add: H=%d and R=%d must be multiples of 4", H, R);
  ZK_CHECK_ARG(dtable != nullptr && payload != nullptr, "zk_rows_scatter_add: null pointer");
  if (R == 0) return 0;
  int g = (R + 3) / 4;
  if (g > 2048) g = 2048;
  const int* ids = reinterpret_cast<const int*>(payload);
  const void* rows = reinterpret_cast<const unsigned char*>(payload) + (size_t)R * 4;
  if (in_bf16)
    hipLaunchKernelGGL(k_rows_scatter_add<true>, dim3(g), dim3(256), 0, stream, dtable, ids, rows, R, H, vocab_rows);
  else
    hipLaunchKernelGGL(k_rows_scatter_add<false>, dim3(g), dim3(256), 0, stream, dtable, ids, rows, R, H, vocab_rows);
  ZK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
