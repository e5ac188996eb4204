// dct_matrix.hpp -- the HEVC core transform matrix, generated at compile time (shared by transform.hip and intra_frame.hip).
// The reference spells it out as literals (32x32 at residual_decode.c:254-288 and :793-827; the smaller sizes are its even rows, truncated).
#pragma once

namespace hevc {

struct Matrix32 { signed char m[32][32]; };

constexpr Matrix32 make_matrix32()
{
	// 64 * sqrt(2) * cos(a * pi / 64) as HEVC rounds it; row 0 of the matrix is the constant 64
	constexpr int mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
		61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
	Matrix32 t = {};
	for (int k = 0; k < 32; ++k)
		for (int i = 0; i < 32; ++i)
		{
			int a = (k * (2 * i + 1)) & 127;
			if (a > 64) a = 128 - a;
			t.m[k][i] = (signed char)(k == 0 ? 64 : (a > 32 ? -mag[64 - a] : mag[a]));
		}
	return t;
}

static constexpr Matrix32 kDct = make_matrix32();

}  // namespace hevc
