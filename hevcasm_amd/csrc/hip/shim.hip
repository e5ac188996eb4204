// shim.hip -- runtime side of the C-ABI: device discovery, error text, memory/stream helpers.
//
// No CPU fallback lives here or anywhere in the library: when no gfx950 device is usable every
// kernel entry point returns HEVCASM_HIP_ENODEV.

#include "common.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace hevc {

static thread_local char g_error[256] = "";

int set_error(int code, const char *what)
{
	std::snprintf(g_error, sizeof(g_error), "hevcasm_hip error %d: %s", code, what ? what : "");
	return code;
}

static bool is_gfx950(int dev)
{
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
	return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

// -1 unknown, else number of gfx950 devices
static std::atomic<int> g_count{-1};

static int count_devices()
{
	int c = g_count.load(std::memory_order_acquire);
	if (c >= 0) return c;
	int n = 0, good = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
	(void)hipGetLastError();
	for (int d = 0; d < n; ++d) good += is_gfx950(d) ? 1 : 0;
	g_count.store(good, std::memory_order_release);
	return good;
}

// per device: -1 unknown, 0 not gfx950, 1 gfx950
static constexpr int kMaxDevices = 64;
static std::atomic<signed char> g_is_gfx950[kMaxDevices];
static std::atomic<bool> g_is_init{false};

static bool device_is_gfx950(int dev)
{
	if (dev < 0 || dev >= kMaxDevices) return is_gfx950(dev);
	if (!g_is_init.load(std::memory_order_acquire))
	{
		for (int d = 0; d < kMaxDevices; ++d) g_is_gfx950[d].store(-1, std::memory_order_relaxed);
		g_is_init.store(true, std::memory_order_release);   // racing initialisers all write the same values
	}
	int v = g_is_gfx950[dev].load(std::memory_order_acquire);
	if (v < 0)
	{
		v = is_gfx950(dev) ? 1 : 0;
		g_is_gfx950[dev].store((signed char)v, std::memory_order_release);
	}
	return v == 1;
}

// Called at the top of every entry point.  The CURRENT device of the calling thread must be a gfx950 (the code object holds
// no other architecture).  Also drops whatever non-sticky error an earlier HIP call of this thread -- ours, the
// application's or another library's -- left behind, so that launch_status() reports this call's launch and nothing else.
int check_device()
{
	if (count_devices() <= 0) return set_error(HEVCASM_HIP_ENODEV, "no usable gfx950 (MI355X) device");
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || !device_is_gfx950(dev))
	{
		(void)hipGetLastError();
		return set_error(HEVCASM_HIP_ENODEV, "the current device is not a gfx950 (MI355X): select one with hevcasm_hip_set_device()");
	}
	(void)hipGetLastError();
	return 0;
}

// compute units of the calling thread's current device (256 on an MI355X in SPX mode, 32 / 128 when the part is partitioned CPX / DPX);
// cached per device; 256 if the query fails
int device_cus()
{
	static std::atomic<int> cus[kMaxDevices];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
	int v = cus[dev].load(std::memory_order_acquire);
	if (v <= 0)
	{
		hipDeviceProp_t p;
		v = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
		cus[dev].store(v, std::memory_order_release);
	}
	return v;
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_device_count(void) { return count_devices(); }

extern "C" int hevcasm_hip_available(void)
{
	if (count_devices() <= 0) return 0;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return 0;
	return device_is_gfx950(dev) ? 1 : 0;
}

extern "C" int hevcasm_hip_set_device(int device)
{
	if (count_devices() <= 0) return set_error(HEVCASM_HIP_ENODEV, "no usable gfx950 (MI355X) device");
	if (!device_is_gfx950(device)) return set_error(HEVCASM_HIP_ENODEV, "that device is not a gfx950 (MI355X)");
	HEVC_HIP_TRY(hipSetDevice(device));
	return 0;
}

extern "C" const char *hevcasm_hip_last_error(void) { return g_error; }

// ---- the per-block path's latch (round 6) ---------------------------------------------------------------------------------
// The reference's per-block function types have no error return (sad.h:50 ...), so a HIP failure behind a HEVCASM_HIP table slot cannot be
// reported by the call that hit it.  What the library can do is make sure it happens ONCE: the first failure latches the device "unavailable"
// for the whole process, with its text, and from then on hevcasm_populate_*() treats the HEVCASM_HIP bit as not set -- the reference's own
// convention for "not available" is a slot filled from the other bits of the mask, or NULL (ssd.c:65-69).  Never a CPU result from a HIP slot.
static std::atomic<int> g_latched{0};
static std::mutex g_latch_mutex;
static char g_latch_text[256] = "";
static hevcasm_hip_failure_handler *g_failure_handler = nullptr;

extern "C" int hevcasm_hip_latched(void) { return g_latched.load(std::memory_order_acquire); }

extern "C" const char *hevcasm_hip_latch_reason(void) { return g_latch_text; }

extern "C" void hevcasm_hip1_latch(const char *reason)
{
	std::lock_guard<std::mutex> lock(g_latch_mutex);
	if (g_latched.load(std::memory_order_relaxed)) return;      // the first failure's text is the one kept
	std::snprintf(g_latch_text, sizeof(g_latch_text), "%s", reason ? reason : "");
	g_latched.store(1, std::memory_order_release);
}

extern "C" void hevcasm_hip_latch_reset(void)
{
	std::lock_guard<std::mutex> lock(g_latch_mutex);
	g_latch_text[0] = 0;
	g_latched.store(0, std::memory_order_release);
}

extern "C" hevcasm_hip_failure_handler *hevcasm_hip_set_failure_handler(hevcasm_hip_failure_handler *handler)
{
	std::lock_guard<std::mutex> lock(g_latch_mutex);
	hevcasm_hip_failure_handler *old = g_failure_handler;
	g_failure_handler = handler;
	return old;
}

// A per-block entry failed: latch, then hand the text to the application's handler; without one, print it and abort (the call's result would be
// undefined and no caller of a function type without an error return checks for that).
extern "C" void hevcasm_hip1_failed(const char *text)
{
	hevcasm_hip1_latch(text);
	hevcasm_hip_failure_handler *h;
	{
		std::lock_guard<std::mutex> lock(g_latch_mutex);
		h = g_failure_handler;
	}
	if (h) { h(text); return; }
	std::fprintf(stderr, "%s\n", text);
	std::abort();
}

// What hevcasm_populate_*() does with the mask it is given: the HEVCASM_HIP bit (1 << 9) survives only while a gfx950 device is usable and no
// per-block call has failed; asking for it without a device latches too, so that the reason can be read back.
extern "C" int hevcasm_hip1_effective_mask(int mask)
{
	const int hip_bit = 1 << 9;
	if (!(mask & hip_bit)) return mask;
	if (hevcasm_hip_latched()) return mask & ~hip_bit;
	if (!hevcasm_hip_available())
	{
		hevcasm_hip1_latch("HEVCASM_HIP asked for, but no usable gfx950 (MI355X) device is current: table slots filled from the other bits of the mask");
		return mask & ~hip_bit;
	}
	return mask;
}

extern "C" int hevcasm_hip_malloc(void **dev_ptr, size_t bytes)
{
	if (int e = check_device()) return e;
	if (!dev_ptr) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	HEVC_HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
	return 0;
}

extern "C" int hevcasm_hip_free(void *dev_ptr)
{
	if (int e = check_device()) return e;
	HEVC_HIP_TRY(hipFree(dev_ptr));
	return 0;
}

extern "C" int hevcasm_hip_upload(void *dev_dst, const void *host_src, size_t bytes, void *stream)
{
	if (int e = check_device()) return e;
	HEVC_HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
	return 0;
}

extern "C" int hevcasm_hip_download(void *host_dst, const void *dev_src, size_t bytes, void *stream)
{
	if (int e = check_device()) return e;
	HEVC_HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
	return 0;
}

extern "C" int hevcasm_hip_stream_create(void **stream)
{
	if (int e = check_device()) return e;
	if (!stream) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t s;
	HEVC_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	*stream = (void *)s;
	return 0;
}

extern "C" int hevcasm_hip_stream_destroy(void *stream)
{
	if (int e = check_device()) return e;
	HEVC_HIP_TRY(hipStreamDestroy(as_stream(stream)));
	return 0;
}

extern "C" int hevcasm_hip_stream_synchronize(void *stream)
{
	if (int e = check_device()) return e;
	HEVC_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
	return 0;
}
