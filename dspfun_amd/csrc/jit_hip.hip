// jit_hip.hip -- plan-time compilation of specialised kernels (jit_kernels.h) with hiprtc, for frame sizes without an entry in
// spec_list.h.  One program per (RowSpecT | ColSpecT) instance holds both transform kinds; it is compiled once per process.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <glob.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "backend.h"

namespace dspfft {

namespace {
enum { JIT_MAXFN = 5 };
struct Built { hipModule_t mod; hipFunction_t fn[JIT_MAXFN]; int nfn; };
std::mutex g_mu;
std::map<std::string, Built> g_cache;        // key: device ordinal | spec type name (a module and its functions belong to the device
                                             // that was current when it was loaded)
// gcnArchName of the current device without its feature suffixes ("gfx950:sramecc+:xnack-" -> "gfx950")
std::string device_arch(int dev)
{
	hipDeviceProp_t pr;
	if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return "gfx950";
	std::string a = pr.gcnArchName;
	const size_t c = a.find(':');
	if (c != std::string::npos) a.resize(c);
	return a.empty() ? "gfx950" : a;
}
// FNV-1a of the headers the program is compiled from: a rebuilt or edited tree never runs a stale code object
unsigned long long source_hash(const char *incdir)
{
	unsigned long long h = 1469598103934665603ull;
	for (const char *name : {"jit_kernels.h", "dct_spec.h", "dct_core.h", "radix.h", "spec_fused.h", "motion_filter.h", "elementwise_core.h", "backend.h"}) {
		FILE *f = fopen((std::string(incdir) + "/" + name).c_str(), "rb");
		if (!f) { h ^= 0xff; h *= 1099511628211ull; continue; }
		unsigned char buf[65536];
		size_t n;
		while ((n = fread(buf, 1, sizeof buf, f)) > 0) for (size_t i = 0; i < n; i++) { h ^= buf[i]; h *= 1099511628211ull; }
		fclose(f);
	}
	return h;
}

std::string clang_resource_include()
{
	const char *root = getenv("ROCM_PATH");
	const std::string pat = std::string(root && *root ? root : "/opt/rocm") + "/lib/llvm/lib/clang/*/include";
	glob_t g;
	std::string r;
	if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc > 0) r = g.gl_pathv[g.gl_pathc - 1];
	globfree(&g);
	return r;
}
// Disk cache of compiled code objects: $DSPFFT_JIT_CACHE, else $XDG_CACHE_HOME/dspfft-jit, else $HOME/.cache/dspfft-jit (created 0700).
// The full key -- spec, kernel names, architecture, hiprtc version, compile options, hash of the kernel headers -- is stored IN the
// file and compared on load, so a stale, foreign or renamed file is recompiled over, never run.
std::string cache_dir()
{
	const char *e = getenv("DSPFFT_JIT_CACHE");
	std::string d;
	if (e && *e) d = e;
	else if ((e = getenv("XDG_CACHE_HOME")) && *e) d = std::string(e) + "/dspfft-jit";
	else if ((e = getenv("HOME")) && *e) d = std::string(e) + "/.cache/dspfft-jit";
	else return "";
	for (size_t i = 1; i <= d.size(); i++) if (i == d.size() || d[i] == '/') { const std::string p = d.substr(0, i); mkdir(p.c_str(), 0700); }
	return d;
}
std::string cache_file(const std::string &key)
{
	const std::string d = cache_dir();
	if (d.empty()) return "";
	unsigned long long h = 1469598103934665603ull;
	for (unsigned char ch : key) { h ^= ch; h *= 1099511628211ull; }
	char name[64];
	snprintf(name, sizeof name, "/%016llx.co", h);
	return d + name;
}
// file: "DSPJIT3\n", the full key, "\n", number of kernels, "\n", their lowered names one per line, code object
bool cache_load(const std::string &path, const std::string &key, std::vector<std::string> &lowered, std::vector<char> &code)
{
	FILE *f = path.empty() ? nullptr : fopen(path.c_str(), "rb");
	if (!f) return false;
	std::vector<char> all;
	char buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + n);
	fclose(f);
	const char *p = all.data(), *end = p + all.size();
	auto line = [&](std::string &out) { const char *q = (const char *)memchr(p, '\n', (size_t)(end - p)); if (!q) return false; out.assign(p, q); p = q + 1; return true; };
	std::string magic, fkey, cnt;
	if (!line(magic) || magic != "DSPJIT3" || !line(fkey) || fkey != key || !line(cnt)) return false;
	const int nk = atoi(cnt.c_str());
	if (nk < 1 || nk > JIT_MAXFN) return false;
	lowered.resize((size_t)nk);
	for (int i = 0; i < nk; i++) if (!line(lowered[(size_t)i])) return false;
	if (p >= end) return false;
	code.assign(p, end);
	return true;
}
void cache_store(const std::string &path, const std::string &key, const std::vector<std::string> &lowered, const std::vector<char> &code)
{
	if (path.empty()) return;
	const std::string tmp = path + "." + std::to_string((long long)getpid());
	FILE *f = fopen(tmp.c_str(), "wb");
	if (!f) return;
	fprintf(f, "DSPJIT3\n%s\n%d\n", key.c_str(), (int)lowered.size());
	for (const std::string &l : lowered) fprintf(f, "%s\n", l.c_str());
	const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
	fclose(f);
	if (ok) rename(tmp.c_str(), path.c_str()); else unlink(tmp.c_str());
}
}  // namespace

bool be_jit_available() { return true; }

// spec_type: e.g. "RowSpecT<float, 1000, 3, 128, 5, 10, 10>".  kind 0 (row): funcs = {REDFT10, REDFT01} and, with `extras`, the 8-bit
// variants {u8 REDFT10, u8 REDFT01}; kind 1 (column): {REDFT10, REDFT01} and, with `extras`, the fused roundtrip.  Returns the number of
// kernels written to funcs (2, 3 or 4) or a negative error.
int be_jit_build(const char *spec_type, int is_col, int extras, const char *incdir, void **funcs, char *log, size_t loglen)
{
	std::lock_guard<std::mutex> lock(g_mu);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return -1;
	const std::string key = std::to_string(dev) + "|" + spec_type + (extras ? "+x" : "");
	auto it = g_cache.find(key);
	if (it != g_cache.end()) { for (int i = 0; i < it->second.nfn; i++) funcs[i] = it->second.fn[i]; return it->second.nfn; }
	std::vector<std::string> names;
	const std::string st = std::string("dspfft::") + spec_type;
	for (int k = 0; k < 2; k++) names.push_back(std::string("dspfft::") + (is_col ? "jit_col" : "jit_row") + "<" + st + ", " + std::to_string(k) + ">");
	if (extras && is_col) names.push_back("dspfft::jit_col_rt<" + st + ">");
	if (extras && !is_col) for (int k = 0; k < 2; k++) names.push_back("dspfft::jit_row_u8<" + st + ", " + std::to_string(k) + ">");
	Built b;
	b.nfn = (int)names.size();
	const std::string arch = device_arch(dev), archopt = "--offload-arch=" + arch;
	int rtc_major = 0, rtc_minor = 0;
	(void)hiprtcVersion(&rtc_major, &rtc_minor);
	char hx[32];
	snprintf(hx, sizeof hx, "%016llx", source_hash(incdir));
	// everything the code object depends on (no newline in it: it is one line of the cache file)
	std::string dkey = std::string(spec_type) + (extras ? "+x" : "") + "|" + arch + "|hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) +
	                   "|-O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize|src " + hx;
	for (const std::string &n : names) dkey += "|" + n;
	const std::string cpath = cache_file(dkey);
	{
		std::vector<std::string> lowered;
		std::vector<char> code;
		if (cache_load(cpath, dkey, lowered, code) && (int)lowered.size() == b.nfn && hipModuleLoadData(&b.mod, code.data()) == hipSuccess) {
			bool ok = true;
			for (int i = 0; i < b.nfn && ok; i++) ok = hipModuleGetFunction(&b.fn[i], b.mod, lowered[(size_t)i].c_str()) == hipSuccess;
			if (ok) {
				g_cache[key] = b;
				for (int i = 0; i < b.nfn; i++) funcs[i] = b.fn[i];
				return b.nfn;
			}
			(void)hipModuleUnload(b.mod);
		}
	}
	std::string src = "#include \"jit_kernels.h\"\n";
	hiprtcProgram prog;
	if (hiprtcCreateProgram(&prog, src.c_str(), "dspfft_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return -1;
	for (const std::string &n : names) hiprtcAddNameExpression(prog, n.c_str());
	const char *root = getenv("ROCM_PATH");
	const std::string inc1 = std::string("-I") + incdir, inc2 = std::string("-I") + (root && *root ? root : "/opt/rocm") + "/include", inc3 = "-I" + clang_resource_include();
	const char *opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=on", "-fno-slp-vectorize", inc1.c_str(), inc2.c_str(), inc3.c_str()};
	const hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
	if (rc != HIPRTC_SUCCESS) {
		size_t ls = 0;
		hiprtcGetProgramLogSize(prog, &ls);
		std::string l(ls, 0);
		if (ls) hiprtcGetProgramLog(prog, &l[0]);
		if (log && loglen) snprintf(log, loglen, "%s", l.c_str());
		hiprtcDestroyProgram(&prog);
		return -2;
	}
	size_t cs = 0;
	hiprtcGetCodeSize(prog, &cs);
	std::vector<char> code(cs);
	hiprtcGetCode(prog, code.data());
	if (hipModuleLoadData(&b.mod, code.data()) != hipSuccess) { hiprtcDestroyProgram(&prog); return -3; }
	std::vector<std::string> low;
	for (int k = 0; k < b.nfn; k++) {
		const char *lowered = nullptr;
		if (hiprtcGetLoweredName(prog, names[(size_t)k].c_str(), &lowered) != HIPRTC_SUCCESS || hipModuleGetFunction(&b.fn[k], b.mod, lowered) != hipSuccess) {
			hiprtcDestroyProgram(&prog);
			(void)hipModuleUnload(b.mod);
			return -4;
		}
		low.push_back(lowered);
	}
	hiprtcDestroyProgram(&prog);
	cache_store(cpath, dkey, low, code);
	g_cache[key] = b;
	for (int i = 0; i < b.nfn; i++) funcs[i] = b.fn[i];
	return b.nfn;
}

// args: one pointer per kernel parameter
int be_jit_launch_n(void *func, void **args, int nwg, int nthr, void *stream)
{
	const hipError_t e = hipModuleLaunchKernel((hipFunction_t)func, (unsigned)nwg, 1, 1, (unsigned)nthr, 1, 1, 0, (hipStream_t)stream, args, nullptr);
	return e == hipSuccess ? 0 : (int)e;
}
int be_jit_launch(void *func, const void *args, int nwg, int nthr, void *stream)
{
	void *params[1] = {const_cast<void *>(args)};
	const hipError_t e = hipModuleLaunchKernel((hipFunction_t)func, (unsigned)nwg, 1, 1, (unsigned)nthr, 1, 1, 0, (hipStream_t)stream, params, nullptr);
	return e == hipSuccess ? 0 : (int)e;
}

}  // namespace dspfft
