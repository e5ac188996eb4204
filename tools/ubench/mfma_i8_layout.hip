// Which k does byte j of lane l carry in the A / B operands of v_mfma_i32_32x32x32_i8 (gfx950)?
// Exact integer check of candidate maps against a CPU product; also checks the C/D map
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_i8_layout.hip -o /tmp/mfma_i8_layout && /tmp/mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const v4i *a, const v4i *b, v16i *d)
{
	v16i acc = {0};
	acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
	d[threadIdx.x] = acc;
}

static int kmap(int cand, int h, int j)
{
	switch (cand)
	{
	case 0: return 16 * h + j;                          // 16 consecutive k per lane half
	case 1: return 8 * h + (j & 7) + 16 * (j >> 3);      // two K = 16 steps back to back
	default: return 4 * h + (j & 3) + 8 * (j >> 2);      // interleaved by dwords
	}
}

int main()
{
	int8_t A[32][32], B[32][32];
	srand(1);
	for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { A[i][j] = (int8_t)(rand() % 255 - 127); B[i][j] = (int8_t)(rand() % 255 - 127); }
	int ref[32][32];
	for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += A[m][kk] * B[kk][n]; ref[m][n] = s; }
	v4i *da, *db; v16i *dd;
	hipMalloc(&da, 64 * sizeof(v4i)); hipMalloc(&db, 64 * sizeof(v4i)); hipMalloc(&dd, 64 * sizeof(v16i));
	for (int cand = 0; cand < 3; ++cand)
	{
		int8_t ha[64][16], hb[64][16];
		for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j)
		{
			const int r = l & 31, h = l >> 5, kk = kmap(cand, h, j);
			ha[l][j] = A[r][kk];
			hb[l][j] = B[kk][r];
		}
		hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
		int hd[64][16];
		hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
		int bad = 0;
		for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 16; ++reg)
		{
			const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
			if (hd[l][reg] != ref[row][col]) ++bad;
		}
		printf("candidate %d: %d of 1024 outputs wrong\n", cand, bad);
	}
	return 0;
}
