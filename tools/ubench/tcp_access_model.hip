// How does the vector L1 (TCP) count "cache accesses" for the load shapes the SAD kernels use?
// Each kernel issues ONE global load per lane from a chosen address pattern; run under
//   rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -- /tmp/tcp_access_model
// and divide the counter by the number of wave-instructions (printed).  Patterns (per wave of 64 lanes):
//   rows x lanes-per-row x bytes-per-lane, row pitch 2048 bytes, first byte of each row at `mis`.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/tcp_access_model.hip -o /tmp/tcp_access_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U1 { uint32_t v; };
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
struct __attribute__((packed, aligned(1))) U3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) U4 { u32x4 v; };

// PITCH = distance between the lanes of a row: BYTES for disjoint lanes, less for lanes whose loads overlap
template <int BYTES, int LPR, int MIS, int PITCH>
__global__ __launch_bounds__(256) void probe(const uint8_t *buf, uint32_t *out, size_t span)
{
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63, row = lane / LPR, c = lane % LPR;
	// every wave reads its own 64 KiB region (cold lines), rows 2048 bytes apart
	const uint8_t *p = buf + (wave * 65536) % span + (size_t)row * 2048 + MIS + c * PITCH;
	uint32_t acc = 0;
	if (BYTES == 4) acc = reinterpret_cast<const U1 *>(p)->v;
	else if (BYTES == 12) { const U3 v = *reinterpret_cast<const U3 *>(p); acc = v.a ^ v.b ^ v.c; }
	else if (BYTES == 8) { const u32x2 v = reinterpret_cast<const U2 *>(p)->v; acc = v.x ^ v.y; }
	else { const u32x4 v = reinterpret_cast<const U4 *>(p)->v; acc = v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES, int LPR, int MIS, int PITCH = BYTES> void run(const uint8_t *buf, uint32_t *out, size_t span)
{
	const int wgs = 4096;
	hipLaunchKernelGGL((probe<BYTES, LPR, MIS, PITCH>), dim3(wgs), dim3(256), 0, 0, buf, out, span);
	hipDeviceSynchronize();
	printf("probe<%2d bytes/lane, %2d lanes/row at pitch %d (%2d rows/wave), misaligned by %d>: %d wave-instructions\n",
		BYTES, LPR, PITCH, 64 / LPR, MIS, wgs * 4);
}

int main()
{
	const size_t span = (size_t)1 << 30;
	uint8_t *buf; uint32_t *out;
	hipMalloc(&buf, span + (1 << 20)); hipMalloc(&out, 4);
	hipMemset(buf, 1, span + (1 << 20));
	// the kernels appear in the profile in this order (distinct template arguments -> distinct names)
	run<16, 64, 0>(buf, out, span); run<16, 64, 1>(buf, out, span); run<16, 64, 4>(buf, out, span);   // one 1 KiB row
	run<16, 4, 0>(buf, out, span); run<16, 4, 1>(buf, out, span); run<16, 4, 4>(buf, out, span); run<16, 4, 20>(buf, out, span); run<16, 4, 68>(buf, out, span);   // 64-byte rows
	run<8, 2, 0>(buf, out, span); run<8, 2, 1>(buf, out, span); run<8, 2, 4>(buf, out, span);   // 16-byte rows
	run<16, 1, 0>(buf, out, span); run<16, 1, 1>(buf, out, span); run<16, 1, 4>(buf, out, span);   // 16-byte rows, one lane
	run<4, 16, 0>(buf, out, span); run<4, 16, 1>(buf, out, span);   // 64-byte rows by dwords
	run<4, 1, 0>(buf, out, span); run<4, 1, 1>(buf, out, span);     // the tail load: one dword per row
	run<16, 2, 0>(buf, out, span); run<16, 2, 1>(buf, out, span); run<16, 2, 4>(buf, out, span);   // 32-byte rows
	// dword-aligned loads wider than the lane's share of the row, so that no tail load is needed: disjoint 12-byte lanes,
	// and lanes whose loads overlap (8-byte pitch with 12-byte loads, 4-byte pitch with 8-byte loads)
	run<12, 2, 0, 12>(buf, out, span); run<12, 2, 4, 12>(buf, out, span);
	run<12, 2, 0, 8>(buf, out, span); run<12, 2, 4, 8>(buf, out, span); run<12, 2, 1, 8>(buf, out, span);
	run<12, 3, 0, 8>(buf, out, span); run<12, 4, 0, 8>(buf, out, span);
	run<8, 3, 0, 4>(buf, out, span); run<8, 2, 0, 4>(buf, out, span);
	run<8, 1, 0>(buf, out, span); run<8, 1, 1>(buf, out, span); run<12, 1, 0>(buf, out, span); run<12, 1, 4>(buf, out, span);
	run<16, 2, 0, 8>(buf, out, span); run<16, 4, 0, 12>(buf, out, span);
	return 0;
}
