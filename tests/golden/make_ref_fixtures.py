"""Regenerates tests/golden/ref_direct.npz from the REFERENCE'S OWN direct-sum code.  Build container only
(needs /root/reference and gcc); never on the GPU box.

FFTW itself is not in /root/reference and was never run.  What the reference does hold is its own
direct-sum statement of the same transforms, and this script compiles that code AS IT LIES:

  scan/scan.c:20-41          generate_basis_matrix + pruned_idct  -- the code scan switches to instead of
                             fftw(execute) on a speed heuristic (scan.c:349-350,446-449): the reference's own
                             definition of the 2-D REDFT01 (summed over ALL coefficients it IS the transform)
  zoom/zoom.c:22-26,34,36-68 scaling_type, min(), generate_scaled_basis
  zoom/zoom.c:361-375        the separable basis x coefficient product (scale 1, offset 0 == REDFT01 / (4wh))
  applybasis/applybasis.c:77-140   the twelve basis functions (dct2 == REDFT10's kernel, dct3 == REDFT01's)
  applybasis/applybasis.c:146-147,370-380,410-425  coords/offsets, the forward / --inverse index aliasing, the
                             partial-sum loops (with the --offset handling of :419-421)
  scan/scan_methods.c:210-228  init_random: the `random` scan order as this image's libc rand() draws it
  scan/scan_methods.c:5-7,11-14,16-184,203-331 + scan/scan_precomputed.c   every scan method that needs no libavutil (all but evalxy /
                             evali): limits, intervals, coordinate generators, radial / iradial / magnitude  ->  tests/golden/ref_scan.npz

  motion/motion.c:18-35,58-60  the spectrogram / preserve-dc enums and struct coords
  motion/motion.c:559-573    scalefactor, normalization, quantizer, threshold per component
  motion/motion.c:644-647,650,683-744,748-751,756-776   motion's elementwise stages around its transforms: uniform-range scaling, the six-face
                             damp / boost, threshold, DC preservation, the quantiser (with its count of coded coefficients), the reverse scaling,
                             and the 8-bit store (clamp + lround)  ->  tests/golden/ref_motion.npz  (COEFF_PRECISION=F, INTERMEDIATE_PRECISION=L as
                             motion/Makefile:1-2 builds; WITHOUT that Makefile's -ffast-math, under which the quantiser's division may become a
                             multiplication by a reciprocal -- what the compiler does then is not the reference's text)

  spec/spec.h:18,22-69, spec/spec.c:66-139, spec/ispec.c:66-67,84-87,92-95,98-163   spec's own encode (DC, uniform range, gain presets incl. `reference`,
                             range one / dc / dcs, log / linear, sign abs / shift / saturate / retain) and ispec's decode (with -p and the sign-map
                             loop), COEFF_PRECISION=F, INTERMEDIATE_PRECISION=D  ->  tests/golden/ref_spec.npz
  motion/motion.c:617-638,755-776 with every --ispec / --spec mode and float_pixels  ->  the io* arrays of tests/golden/ref_motion.npz

The text of those line ranges is read from /root/reference at generation time into a temporary translation unit
that includes the reference's include/precision.h (COEFF_PRECISION=L, INTERMEDIATE_PRECISION=L: the tightest build
the reference offers) and is compiled with plain gcc -- no stand-in headers; no reference text is written to the
repository.  Only numbers (inputs and outputs) go into the fixture file.
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DSPFUN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import synth_f32, splitmix64_stream  # noqa: E402

LD = np.longdouble
FUNCS = ["dft", "idft", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "wht", "dht"]


def lines(path, a, b):
    with open(os.path.join(REF, path)) as f:
        src = f.read().split("\n")
    return "\n".join(src[a - 1:b]) + "\n"


def build_scan_zoom(tmp):
    """scan.c:20-41 and zoom.c's basis + product in one TU (both only need precision.h and libm)."""
    tu = "#include <stdlib.h>\n#include <stdbool.h>\n#include <math.h>\n#include \"precision.h\"\n"
    tu += lines("scan/scan.c", 20, 41)
    tu += lines("zoom/zoom.c", 22, 26) + lines("zoom/zoom.c", 34, 34) + lines("zoom/zoom.c", 36, 68)
    # --- wrappers (this script's own code) -------------------------------------------------------------
    tu += """
void ref_pruned_idct(const long double *c, long double *img, const size_t *coords, size_t ncoords, size_t width, size_t height, size_t channels)
{   /* scan.c:357-362,449 */
	coeff *basis[2];
	basis[0] = generate_basis_matrix(height);
	basis[1] = width == height ? basis[0] : generate_basis_matrix(width);
	pruned_idct(basis, (coeff *)c, img, (size_t (*)[2])coords, ncoords, width, height, channels);
	free(basis[0]);
	if (width != height) free(basis[1]);
}
size_t ref_zoom(const long double *coeffs, long double *icoeffs, int scaling_type_, long double xscale_num, long double xscale_den, long double yscale_num,
                long double yscale_den, long double vx, long double vy, size_t vw, size_t vh, size_t width, size_t height, size_t *out_cheight)
{   /* zoom.c:323-325,347-359 */
	enum scaling_type scaling_type = scaling_type_;
	coeff *xbasis = NULL, *ybuf = NULL;
	intermediate *tmp = NULL;
	size_t maxvectors = vh > vw ? vh : vw;
	bool reuse_basis = width == height && vx == vy && xscale_num == yscale_num && xscale_den == yscale_den;
	size_t cwidth = generate_scaled_basis(&xbasis,scaling_type,xscale_num,xscale_den,vx,(reuse_basis ? maxvectors : vw),width);
	size_t cheight;
	coeff* ybasis;
	if(reuse_basis) {
		cheight = cwidth;
		ybasis = xbasis;
	}
	else {
		cheight = generate_scaled_basis(&ybuf,scaling_type,yscale_num,yscale_den,vy,vh,height);
		ybasis = ybuf;
	}
	tmp = realloc(tmp,sizeof(*tmp)*cheight);
"""
    tu += lines("zoom/zoom.c", 361, 375)
    tu += """
	free(xbasis); free(ybuf); free(tmp);
	*out_cheight = cheight;
	return cwidth;
}
"""
    return compile_tu(tmp, "scanzoom", tu)


def build_applybasis(tmp):
    tu = "#include <stdlib.h>\n#include <string.h>\n#include <stdbool.h>\n#include <complex.h>\n#include <math.h>\n#include \"precision.h\"\n"
    tu += "#ifndef I\n#define I _Complex_I\n#endif\n"
    tu += lines("applybasis/applybasis.c", 77, 140)
    tu += lines("applybasis/applybasis.c", 146, 147)
    tu += "static complex_intermediate (*const table[12])(long long, long long, unsigned long long, bool) = {" + ",".join(FUNCS) + "};\n"
    tu += """
void ref_basis(int f, long long k, long long n, unsigned long long N, int ortho, long double *re, long double *im)
{
	complex_intermediate v = table[f](k, n, N, ortho);
	*re = creall(v); *im = cimagl(v);
}
/* the `.coeff` file as applybasis.c:381-388 (header: coords dumpsize) and :443 (one fwrite of `complex_intermediate partsums[3]` per term) lay it out,
   with the reference's own types at this build's INTERMEDIATE_PRECISION: vals[term][3][2] = (re, im) */
#include <stdio.h>
int ref_write_coeff(const char *path, unsigned long long w, unsigned long long h, const long double *vals, size_t nterms)
{
	coords dumpsize = {{w, h}};
	FILE *df = fopen(path, "w");
	if (!df || fwrite(&dumpsize, sizeof(dumpsize), 1, df) != 1) return 1;
	for (size_t t = 0; t < nterms; t++) {
		complex_intermediate partsums[3];
		for (int j = 0; j < 3; j++) partsums[j] = vals[(t * 3 + j) * 2] + I * vals[(t * 3 + j) * 2 + 1];
		if (fwrite(partsums, sizeof(partsums), 1, df) != 1) return 1;
	}
	return fclose(df);
}
/* the partial sums of applybasis.c:370-380,410-425; out[term][3] complex, terms in loop order */
void ref_partsums(int f, int orthogonal, int inverse, coords insize, coords terms, coords partsum, offsets offset, const long double *pixels_, long double *out)
{
	complex_intermediate (*function)(long long, long long, unsigned long long, bool) = table[f];
	const intermediate *pixels = pixels_;
	coords size = insize;
	size_t term = 0;
"""
    tu += lines("applybasis/applybasis.c", 370, 380)
    tu += lines("applybasis/applybasis.c", 410, 425)
    tu += """
					for (int j = 0; j < 3; j++) { out[(term * 3 + j) * 2] = creall(partsums[j]); out[(term * 3 + j) * 2 + 1] = cimagl(partsums[j]); }
					term++;
				}
}
"""
    return compile_tu(tmp, "applybasis", tu)


def build_random(tmp):
    """scan_methods.c:210-228 init_random (libc srand / rand) as it lies"""
    tu = "#include <stdlib.h>\n#include <time.h>\n#include \"precision.h\"\n"
    tu += lines("scan/scan_methods.c", 210, 228)
    tu += "size_t *ref_init_random(size_t w, size_t h, const char *args) { return init_random(w, h, 3, NULL, args); }\n"
    return compile_tu(tmp, "random", tu)


def build_scan_methods(tmp):
    """scan/scan_methods.c without its libavutil parts: :5-7 (its own headers), :11-14, :16-184 (limits, intervals, the scan functions
    through scan_mirror), :203-331 (scan_precomputed, init_random, init_magnitude, round_function, init_radial, init_iradial) -- i.e.
    everything except <libavutil/eval.h> (:9), scan_evali (:186-201), init_evalxy / init_evali (:333-391) and the tables that name them.
    Linked with the reference's scan_precomputed.c as it lies; COEFF_PRECISION=F, INTERMEDIATE_PRECISION=D as scan/Makefile:1-2 builds."""
    tu = lines("scan/scan_methods.c", 5, 7) + lines("scan/scan_methods.c", 11, 14) + lines("scan/scan_methods.c", 16, 184) + lines("scan/scan_methods.c", 203, 331)
    tu += """
/* ---- this script's own dispatch over the reference's static functions ---- */
typedef void (*scan_fn)(void*, size_t, size_t, size_t, size_t (*)[2]);
static scan_fn fn_of(int m) { scan_fn t[] = {scan_horiz, scan_vert, scan_zigzag, scan_row, scan_col, scan_diag, scan_mirror, scan_box, scan_ibox}; return t[m]; }
size_t ref_limit(int m, size_t w, size_t h)
{
	switch (m) { case 3: return limit_height(0, w, h); case 4: return limit_width(0, w, h); case 5: return limit_sum(0, w, h); case 6: case 7: return limit_max(0, w, h);
	             case 8: return limit_min(0, w, h); default: return w * h; }          /* scan_context.c:30: limit ? limit() : width * height */
}
size_t ref_max_interval(int m, size_t w, size_t h)
{
	switch (m) { case 3: return limit_width(0, w, h); case 4: return limit_height(0, w, h); case 5: return limit_min(0, w, h); case 6: return limit_mirror(0, w, h);
	             case 7: case 8: return limit_sum(0, w, h); default: return 1; }      /* the .max_interval entries of :453-567; scan_context.c:31 */
}
size_t ref_interval(int m, size_t w, size_t h, size_t i)
{
	switch (m) { case 3: return w; case 4: return h; case 5: return interval_diag(0, w, h, i); case 6: return interval_mirror(0, w, h, i);
	             case 7: return interval_box(0, w, h, i); case 8: return interval_ibox(0, w, h, i); default: return 1; }
}
void ref_scan(int m, size_t w, size_t h, size_t i, size_t (*coords)[2]) { fn_of(m)(0, w, h, i, coords); }
/* FNV-1a-64 over y*w+x of every coordinate in scan order (the word variant of SURVEY 8c / tests/golden/scan_golden.json), looped here in C: one
   index per pixel is 33 M calls at 7680x4320 */
unsigned long long ref_scan_fnv1a(int m, size_t w, size_t h)
{
	unsigned long long hsh = 1469598103934665603ULL;
	size_t lim = ref_limit(m, w, h), mi = ref_max_interval(m, w, h);
	size_t (*buf)[2] = malloc(sizeof(*buf) * (mi + 2));
	for (size_t i = 0; i < lim; i++) {
		size_t n = ref_interval(m, w, h, i);
		fn_of(m)(0, w, h, i, buf);
		for (size_t k = 0; k < n; k++) { hsh ^= (unsigned long long)(buf[k][0] * w + buf[k][1]); hsh *= 1099511628211ULL; }
	}
	free(buf);
	return hsh;
}
/* precomputed methods: 0 radial, 1 iradial, 2 magnitude (coeffs: w*h*channels floats; args as on the command line or NULL) */
struct scan_precomputed *ref_precomputed(int which, size_t w, size_t h, size_t channels, float *coeffs, const char *args)
{
	return which == 0 ? init_radial(w, h, channels, coeffs, args) : which == 1 ? init_iradial(w, h, channels, coeffs, args) : init_magnitude(w, h, channels, coeffs, args);
}
"""
    src = os.path.join(tmp, "scanmethods.c")
    so = os.path.join(tmp, "scanmethods.so")
    with open(src, "w") as f:
        f.write(tu)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-DCOEFF_PRECISION=F", "-DINTERMEDIATE_PRECISION=D", "-O2", "-fPIC", "-shared", "-w",
                           "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "scan"), src, os.path.join(REF, "scan", "scan_precomputed.c"), "-o", so, "-lm"])
    return C.CDLL(so)


def build_motion(tmp):
    """motion.c's elementwise stages as they lie, around this script's own declarations of the variables they use (one component, i = 0)"""
    tu = "#include <stdlib.h>\n#include <stdint.h>\n#include <stdbool.h>\n#include <string.h>\n#include <math.h>\n#include \"precision.h\"\n#include \"keyed_enum.h\"\n"
    tu += lines("motion/motion.c", 18, 35) + lines("motion/motion.c", 58, 60)
    tu += """
/* stages: 1 = uniform-range scaling (:644-647), 2 = damp .. quantiser (:683-744), 4 = reverse scaling (:748-751), 8 = 8-bit store (:756-776) */
unsigned long long ref_motion_stages(int stages, float *coeffs_, unsigned char *pblock_, const uint64_t *active_, const uint64_t *minbuf_, const uint64_t *scaled_,
                                     const uint64_t *block_, const uint64_t *band_begin, const uint64_t *band_end, long double damp_, long double boost_,
                                     long double threshold_min_, long double threshold_max_, int preserve_dc_, long double quant_)
{
	const int components = 1, i = 0;
	coords active = {{active_[0], active_[1], active_[2]}}, minbuf = {{minbuf_[0], minbuf_[1], minbuf_[2]}}, scaled = {{scaled_[0], scaled_[1], scaled_[2]}};
	coords block = {{block_[0], block_[1], block_[2]}};
	range bandpass = {{{band_begin[0], band_begin[1], band_begin[2]}}, {{band_end[0], band_end[1], band_end[2]}}};
	intermediate damp[1] = {damp_}, boost[1] = {boost_}, threshold_min = threshold_min_, threshold_max = threshold_max_, quant = quant_;
	enum spectype spec = spectype_none;
	enum ispectype ispec = ispectype_none;
	enum preserve_dctype preserve_dc = preserve_dc_;
	void *expr = NULL;
	bool float_pixels = false, linear = false, dithering = false;
	intermediate (*output_trc)(intermediate) = NULL;
	coeff *coeffs = coeffs_;
	void *pblock = pblock_;
	uint64_t coeffs_coded = 0;
"""
    tu += lines("motion/motion.c", 559, 573)
    tu += "\tif (stages & 1) {\n" + lines("motion/motion.c", 644, 647) + "\t}\n"
    tu += lines("motion/motion.c", 650, 650)
    tu += "\tif (stages & 2) {\n" + lines("motion/motion.c", 683, 744) + "\t}\n"
    tu += "\tif (stages & 4) {\n" + lines("motion/motion.c", 748, 751) + "\t}\n"
    tu += "\tif (stages & 8) {\n" + lines("motion/motion.c", 756, 776) + "\t\t\t\t\t\t}\n\t}\n"
    tu += "\t(void)c; (void)ic; (void)components; (void)float_pixels; (void)linear; (void)dithering; (void)output_trc; (void)block; (void)dc;\n\treturn coeffs_coded;\n}\n"
    src = os.path.join(tmp, "motion.c")
    so = os.path.join(tmp, "motion.so")
    with open(src, "w") as f:
        f.write(tu)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-DCOEFF_PRECISION=F", "-DINTERMEDIATE_PRECISION=L", "-O2", "-fPIC", "-shared", "-w",
                           "-I" + os.path.join(REF, "include"), src, "-o", so, "-lm"])
    return C.CDLL(so)


def build_motion_io(tmp):
    """motion.c's pixel load (:617-638, with the --ispec decodes and float_pixels) and its output stage (:755-776, with the --spec encodes,
    the constant of `abs` from the block's DC (:755) and float_pixels), as they lie, around this script's own declarations (one component, i = 0)"""
    tu = "#include <stdlib.h>\n#include <stdint.h>\n#include <stdbool.h>\n#include <string.h>\n#include <math.h>\n#include \"precision.h\"\n#include \"keyed_enum.h\"\n"
    tu += lines("motion/motion.c", 18, 35) + lines("motion/motion.c", 58, 60)
    decl = """
	const int components = 1, i = 0;
	coords minbuf = {{minbuf_[0], minbuf_[1], minbuf_[2]}}, scaled = {{scaled_[0], scaled_[1], scaled_[2]}}, block = {{block_[0], block_[1], block_[2]}};
	intermediate threshold_min = 0, threshold_max = 0, quant = 0;
	enum spectype spec = spec_;
	enum ispectype ispec = ispec_;
	bool float_pixels = float_pixels_, linear = false, dithering = false;
	intermediate (*input_trc)(intermediate) = NULL, (*output_trc)(intermediate) = NULL;
	coeff *coeffs = coeffs_;
	void *pblock = pblock_;
	size_t mincomponent = minbuf[0].w * minbuf[0].h * minbuf[0].d;
"""
    tu += """
/* c_ic_out[0] = c[0], [1] = ic[0] (motion.c:568-569) */
void ref_motion_load(float *coeffs_, void *pblock_, const uint64_t *minbuf_, const uint64_t *scaled_, const uint64_t *block_, int ispec_, int float_pixels_, long double *c_ic_out)
{
	const int spec_ = 0;
""" + decl
    tu += lines("motion/motion.c", 559, 573)
    tu += lines("motion/motion.c", 617, 638)
    tu += "\tc_ic_out[0] = c[0]; c_ic_out[1] = ic[0];\n\t(void)components; (void)linear; (void)dithering; (void)input_trc; (void)output_trc; (void)quantizer; (void)threshold; (void)scalefactor; (void)mincomponent;\n}\n"
    tu += """
/* dc_: the block's DC coefficient as :650 takes it before the filters */
void ref_motion_store(float *coeffs_, void *pblock_, const uint64_t *minbuf_, const uint64_t *scaled_, const uint64_t *block_, int spec_, int float_pixels_, float dc_, long double *c_ic_out)
{
	const int ispec_ = 0;
	coeff dc = dc_;
""" + decl
    tu += lines("motion/motion.c", 559, 573)
    tu += "\tif(!spec) {}\n" + lines("motion/motion.c", 755, 776) + "\t\t\t\t\t\t}\n"
    tu += "\tc_ic_out[0] = c[0]; c_ic_out[1] = ic[0];\n\t(void)components; (void)linear; (void)dithering; (void)input_trc; (void)output_trc; (void)quantizer; (void)threshold; (void)mincomponent; (void)ispec;\n}\n"
    src = os.path.join(tmp, "motion_io.c")
    so = os.path.join(tmp, "motion_io.so")
    with open(src, "w") as f:
        f.write(tu)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-DCOEFF_PRECISION=F", "-DINTERMEDIATE_PRECISION=L", "-O2", "-fPIC", "-shared", "-w",
                           "-I" + os.path.join(REF, "include"), src, "-o", so, "-lm"])
    return C.CDLL(so)


ISPEC = {"none": 0, "shift": 1, "flat": 2, "copy": 3}             # enum ispectype (motion.c:24-27 through keyed_enum.h: none first)
SPEC = {"none": 0, "abs": 1, "shift": 2, "flat": 3, "copy": 4}    # enum spectype (motion.c:18-22)


def motion_io_fixtures(tmp, out):
    """io{g}_*: the pixel load and the output stage for every --ispec / --spec mode, 8-bit and float pixels, on blocks embedded in a larger buffer"""
    mo = build_motion_io(tmp)
    vp = C.c_void_p
    mo.ref_motion_load.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    mo.ref_motion_store.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]
    u64 = lambda *v: np.array(v, dtype=np.uint64)
    geoms = [((2, 9, 16), (2, 12, 20)), ((1, 45, 80), (1, 48, 84))]        # (d, h, w) block = scaled inside minbuf (d, h, w)
    out["io_geoms"] = np.array([[*a, *m] for a, m in geoms], dtype=np.int64)
    for g, ((d, h, w), (md, mh, mw)) in enumerate(geoms):
        n = md * mh * mw
        MB, A = u64(mw, mh, md), u64(w, h, d)
        pix8 = (splitmix64_stream(0xD5F1C00 + g, n) >> np.uint64(56)).astype(np.uint8)
        pixf = synth_f32(0xD5F1D00 + g, n)
        out[f"io{g}_pix_u8"] = pix8
        out[f"io{g}_pix_f32"] = pixf
        cic = np.zeros(2, dtype=np.longdouble)
        for name, code in ISPEC.items():
            for fpx, pix in ((0, pix8), (1, pixf)):
                cbuf = np.full(n, np.float32(-77.0))                            # (:617 zeroes the whole buffer first)
                p = pix.copy()
                mo.ref_motion_load(cbuf.ctypes.data, p.ctypes.data, MB.ctypes.data, A.ctypes.data, A.ctypes.data, code, fpx, cic.ctypes.data)
                out[f"io{g}_load_{name}_{'f32' if fpx else 'u8'}"] = cbuf
                if name == "shift":
                    out[f"io{g}_ic"] = np.array([float(cic[1])])
        # coefficients of the size the uniform-range forward transform of 8-bit samples leaves (motion_fixtures), DC the largest
        x = ((synth_f32(0xD5F1E00 + g, n) * 2 - 1) * np.float32(255.0 * np.sqrt(8.0 * w * h * d)) * (synth_f32(0xD5F1F00 + g, n) ** 6)).astype(np.float32)
        x[0] = np.float32(127.0 * np.sqrt(8.0 * w * h * d))
        out[f"io{g}_coeffs"] = x
        for name, code in SPEC.items():
            for fpx in (0, 1):
                pb = np.zeros(n, dtype=np.float32 if fpx else np.uint8)
                cb = x.copy()
                mo.ref_motion_store(cb.ctypes.data, pb.ctypes.data, MB.ctypes.data, A.ctypes.data, A.ctypes.data, code, fpx, float(x[0]), cic.ctypes.data)
                out[f"io{g}_store_{name}_{'f32' if fpx else 'u8'}"] = pb
                if name in ("abs", "shift") and not fpx:
                    out[f"io{g}_c_{name}"] = np.array([float(cic[0])])
        print("motion io geometry", g, (d, h, w), (md, mh, mw))


def build_spec(tmp):
    """spec/spec.c:66-139 (DC, uniform range, gain, range, scale, sign) and spec/ispec.c:66-67,84-87,92-95,98-163 (the decode, with the sign-map
    pixel loop but without the MagickWand calls that read the map image, :88-91,96-97) as they lie, around this script's own declarations;
    enums and option structs from spec/spec.h:18,22-69 by line range (the header itself includes fftw3.h and MagickWand).
    COEFF_PRECISION=F, INTERMEDIATE_PRECISION=D: the build whose types dspfft_spec_encode / dspfft_ispec_decode have (float samples, double scalars)."""
    tu = "#include <stdlib.h>\n#include <stdint.h>\n#include <stdbool.h>\n#include <string.h>\n#include <math.h>\n#include \"precision.h\"\n#include \"keyed_enum.h\"\n"
    tu += lines("spec/spec.h", 18, 18) + lines("spec/spec.h", 22, 69)
    tu += """
/* f_: w*h*d_ raw REDFT10^2 outputs (spec.c:64); normalised_: f after :78; DC_out: :66-68; gain_out: :81-87 */
void ref_spec(float *f_, size_t w, size_t h, size_t d_, int scaletype, int signtype, int gaintype, int rangetype, double custom_gain, float *normalised_, double *DC_out, double *gain_out)
{
	struct specopts opts = {0};
	opts.params = (struct specparams){scaletype, signtype, gaintype, rangetype};
	opts.gain = custom_gain;
	size_t l = w * h * d_, d = d_;
	size_t i, y, z;
	coeff *f = f_;
"""
    tu += lines("spec/spec.c", 66, 78)
    tu += "\tmemcpy(normalised_, f, l * sizeof(coeff)); memcpy(DC_out, DC, d * sizeof(double));\n"
    tu += lines("spec/spec.c", 81, 139)
    tu += "\t*gain_out = gain;\n}\n"
    tu += """
/* f_: the spectrogram samples; DC_: the header's DC terms (overwritten from the map when map_ is given, ispec.c:92-93); map_: 8-bit sign map or NULL */
void ref_ispec(float *f_, size_t w, size_t h, size_t d_, int scaletype, int signtype, int gaintype, int rangetype, double custom_gain, double *DC_, int preserve_dc_, unsigned char *map_, double *gain_out)
{
	struct specopts opts = {0};
	opts.params = (struct specparams){scaletype, signtype, gaintype, rangetype};
	opts.gain = custom_gain;
	size_t l = w * h * d_, d = d_;
	size_t i, y, z;
	coeff *f = f_;
	bool preserve_dc = preserve_dc_;
	const char *signmap = map_ ? "map" : NULL;
"""
    tu += lines("spec/ispec.c", 66, 67)
    tu += "\tmemcpy(DC, DC_, d * sizeof(double));\n"
    tu += lines("spec/ispec.c", 84, 87) + "\t\t\tunsigned char* tmp = map_;\n" + lines("spec/ispec.c", 92, 95) + lines("spec/ispec.c", 98, 163)
    tu += "\t*gain_out = gain; memcpy(DC_, DC, d * sizeof(double));\n}\n"
    src = os.path.join(tmp, "spec.c")
    so = os.path.join(tmp, "spec.so")
    with open(src, "w") as f:
        f.write(tu)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-DCOEFF_PRECISION=F", "-DINTERMEDIATE_PRECISION=D", "-O2", "-fPIC", "-shared", "-w",
                           "-I" + os.path.join(REF, "include"), src, "-o", so, "-lm"])
    return C.CDLL(so)


# spec/spec.h:29-47 through keyed_enum.h (`none` first)
SPEC_SIGN = {"abs": 1, "shift": 2, "saturate": 3, "retain": 4}
SPEC_RANGE = {"one": 1, "dc": 2, "dcs": 3}
SPEC_SCALE = {"linear": 1, "log": 2}
SPEC_GAIN = {"native": 1, "reference": 2, "custom": 3}


def spec_fixtures(tmp):
    """tests/golden/ref_spec.npz: every range x scale x sign combination of spec's encode and ispec's decode (the three gain presets cycled over
    them), one- and three-channel images; ispec additionally with -p (preserve DC) and with a sign map for the `abs` spectrograms"""
    sp = build_spec(tmp)
    vp, st = C.c_void_p, C.c_size_t
    sp.ref_spec.argtypes = [vp, st, st, st, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, vp]
    sp.ref_ispec.argtypes = [vp, st, st, st, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, C.c_int, vp, vp]
    out = {}
    cases = []
    k = 0
    for (h, w, d) in ((9, 16, 3), (12, 10, 1)):
        l = h * w * d
        # raw REDFT10^2 outputs of an image in [0, 1): DC = 4 w h mean, the rest decays with frequency
        img = synth_f32(0xD5F2000 + d, l).reshape(h, w, d).astype(np.float64)
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        raw = (rnd(0xD5F2100 + d, l).reshape(h, w, d) * (4.0 * w * h) / (1.0 + yy + xx)[..., None] ** 1.5)
        raw[0, 0] = 4.0 * w * h * img.mean(axis=(0, 1)) * (1.0 + 0.2 * np.arange(d))
        raw = raw.astype(np.float32)
        for rname, rcode in SPEC_RANGE.items():
            for sname, scode in SPEC_SCALE.items():
                for gname, gcode in SPEC_SIGN.items():
                    gain_name = list(SPEC_GAIN)[k % 3]
                    custom = 96.0 + k
                    f = raw.copy().ravel()
                    normalised = np.zeros(l, dtype=np.float32)
                    DC = np.zeros(d); gain = np.zeros(1)
                    sp.ref_spec(f.ctypes.data, w, h, d, scode, gcode, SPEC_GAIN[gain_name], rcode, custom, normalised.ctypes.data, DC.ctypes.data, gain.ctypes.data)
                    out[f"s{k}_raw"] = raw.ravel().copy(); out[f"s{k}_normalised"] = normalised; out[f"s{k}_encoded"] = f.copy(); out[f"s{k}_dc"] = DC.copy()
                    for pdc in (0, 1):
                        g2 = f.copy(); DC2 = DC.copy(); gain2 = np.zeros(1)
                        sp.ref_ispec(g2.ctypes.data, w, h, d, scode, gcode, SPEC_GAIN[gain_name], rcode, custom, DC2.ctypes.data, pdc, None, gain2.ctypes.data)
                        out[f"s{k}_decoded_p{pdc}"] = g2
                        assert gain2[0] == gain[0]
                    if gname == "abs":
                        # the sign map spec -t sign writes: 8-bit samples of the `saturate` spectrogram (1 -> 255, 0 -> 0), first pixel = the DC terms
                        smap = np.where(normalised >= 0, 255, 0).astype(np.uint8)
                        smap[:d] = np.clip(np.round(DC * 255), 0, 255).astype(np.uint8)
                        g3 = f.copy(); DC3 = DC.copy(); gain3 = np.zeros(1)
                        sp.ref_ispec(g3.ctypes.data, w, h, d, scode, gcode, SPEC_GAIN[gain_name], rcode, custom, DC3.ctypes.data, 1, smap.ctypes.data, gain3.ctypes.data)
                        out[f"s{k}_signmap"] = smap; out[f"s{k}_decoded_signmap"] = g3; out[f"s{k}_dc_signmap"] = DC3.copy()
                    cases.append([h, w, d, rcode, scode, gcode, SPEC_GAIN[gain_name], custom, gain[0]])
                    k += 1
    out["cases"] = np.array(cases, dtype=np.float64)
    path = os.path.join(HERE, "ref_spec.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), len(cases), "cases")


def motion_fixtures(tmp):
    """inputs and outputs of motion's elementwise stages on blocks embedded in a larger buffer; every stage alone and the whole chain"""
    mo = build_motion(tmp)
    vp, ld = C.c_void_p, C.c_longdouble
    mo.ref_motion_stages.restype = C.c_ulonglong
    mo.ref_motion_stages.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, ld, ld, ld, ld, C.c_int, ld]
    u64 = lambda *v: np.array(v, dtype=np.uint64)
    out = {}
    # (d, h, w) active block inside minbuf (d, h, w); band [begin, end) per (d, h, w); damp, boost, thr_min, thr_max, preserve_dc, quant
    cases = [
        ((1, 12, 16), (1, 12, 16), (0, 0, 0), (1, 12, 16), 1.0, 1.0, 0.0, 0.0, 0, 20.0),          # quantiser alone (motion --quant 20)
        ((1, 45, 80), (1, 48, 84), (0, 0, 0), (1, 45, 80), 1.0, 1.0, 0.0, 0.0, 0, 3.0),           # ... in an embedding, another qfactor
        ((4, 10, 12), (4, 12, 16), (1, 2, 3), (3, 8, 9), 0.3, 1.0, 0.0, 0.0, 0, 0.0),             # six-face damp
        ((4, 10, 12), (4, 12, 16), (0, 0, 0), (2, 5, 6), 0.25, 1.75, 0.0, 0.0, 0, 0.0),           # damp outside + boost inside a corner box
        ((2, 9, 16), (2, 9, 16), (0, 0, 0), (2, 9, 16), 1.0, 1.0, 0.002, 0.5, 0, 0.0),            # threshold
        ((3, 8, 8), (3, 8, 8), (0, 1, 0), (3, 8, 8), 0.5, 1.0, 0.0, 0.0, 1, 0.0),                 # dc stop, preserve dc
        ((3, 8, 8), (3, 8, 8), (0, 0, 0), (3, 6, 6), 0.5, 1.5, 0.0, 0.0, 2, 0.0),                 # boost, preserve grey
        ((2, 16, 24), (2, 16, 24), (0, 0, 2), (2, 12, 20), 0.6, 1.2, 0.001, 0.8, 1, 7.0),         # everything at once
    ]
    out["cases"] = np.array([[*a, *m, *b0, *b1, da, bo, t0, t1, pd, q] for (a, m, b0, b1, da, bo, t0, t1, pd, q) in cases], dtype=np.float64)
    for ci, (a, m, b0, b1, da, bo, t0, t1, pd, q) in enumerate(cases):
        d, h, w = a
        md, mh, mw = m
        n = md * mh * mw
        # coefficients of the size a forward transform of 8-bit samples leaves: the uniform-range values are O(255 sqrt(8 whd)) at DC, small elsewhere
        x = ((synth_f32(0xD5F1900 + ci, n) * 2 - 1) * np.float32(255.0 * np.sqrt(8.0 * w * h * d)) * (synth_f32(0xD5F1A00 + ci, n) ** 6)).astype(np.float32)
        x[0] = np.float32(127.0 * np.sqrt(8.0 * w * h * d))
        A, MB, B0, B1 = u64(w, h, d), u64(mw, mh, md), u64(b0[2], b0[1], b0[0]), u64(b1[2], b1[1], b1[0])      # (kept alive across the calls)
        args = lambda stages, buf, pb: mo.ref_motion_stages(stages, buf.ctypes.data, pb.ctypes.data, A.ctypes.data, MB.ctypes.data, A.ctypes.data, A.ctypes.data,
                                                            B0.ctypes.data, B1.ctypes.data, da, bo, t0, t1, pd, q)
        pb = np.zeros(n, dtype=np.uint8)
        out[f"m{ci}_in"] = x
        for stages, name in ((1, "scaled"), (2, "filtered"), (4, "unscaled"), (3, "scaled_filtered"), (7, "chain")):
            b = x.copy()
            coded = args(stages, b, pb)
            out[f"m{ci}_{name}"] = b
            if stages & 2:
                out[f"m{ci}_{name}_coded"] = np.array([coded], dtype=np.uint64)
        # the 8-bit store on values around the whole 0..255 range and beyond, with exact halves among them
        v = ((synth_f32(0xD5F1B00 + ci, n) * 300 - 20) / (np.float64(1.0))).astype(np.float64)
        v[:16] = np.arange(16) + 0.5
        # pel = coeffs * scalefactor * normalization^2 with scalefactor = 1 (scaled == block): coefficients that land on those pels
        norm2 = 1.0 / (8.0 * w * h * d)
        cb = (v / norm2).astype(np.float32)
        pb = np.zeros(n, dtype=np.uint8)
        args(8, cb, pb)
        out[f"m{ci}_store_in"] = cb
        out[f"m{ci}_store_u8"] = pb.copy()
        print("motion case", ci, a, m)
    motion_io_fixtures(tmp, out)
    path = os.path.join(HERE, "ref_motion.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


SCAN_METHODS = ["horizontal", "vertical", "zigzag", "row", "column", "diagonal", "mirror", "box", "ibox"]


def fnv_words(words):
    """order-sensitive 64-bit checksum that numpy can evaluate: sum_k word_k * (2k + 1) mod 2^64"""
    wv = np.asarray(words, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return int((wv * (np.arange(len(wv), dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def scan_fixtures(tmp):
    """every scan method but random / file / evalxy / evali: coordinate lists at small sizes, hashes at real ones"""
    sm = build_scan_methods(tmp)
    st = C.c_size_t
    sm.ref_limit.restype = sm.ref_max_interval.restype = sm.ref_interval.restype = st
    sm.ref_limit.argtypes = sm.ref_max_interval.argtypes = [C.c_int, st, st]
    sm.ref_interval.argtypes = [C.c_int, st, st, st]
    sm.ref_scan.argtypes = [C.c_int, st, st, st, C.c_void_p]

    class Pre(C.Structure):
        _fields_ = [("limit", st), ("intervals", C.POINTER(st)), ("scans", C.POINTER(C.POINTER(st * 2)))]
    sm.ref_precomputed.restype = C.POINTER(Pre)
    sm.ref_precomputed.argtypes = [C.c_int, st, st, st, C.c_void_p, C.c_char_p]
    out = {}
    small = [(8, 8), (16, 9), (9, 16), (33, 20), (1, 7), (7, 1)]
    big = [(1920, 1080), (1080, 1920), (640, 480)]
    out["small_sizes"] = np.array(small); out["big_sizes"] = np.array(big)
    for m, name in enumerate(SCAN_METHODS):
        for (w, h) in small + big:
            lim, mi = sm.ref_limit(m, w, h), sm.ref_max_interval(m, w, h)
            buf = np.zeros((mi + 2, 2), dtype=np.uint64)
            flat, counts = [], []
            for i in range(lim):
                n = sm.ref_interval(m, w, h, i)
                sm.ref_scan(m, w, h, i, buf.ctypes.data)
                counts.append(n)
                flat.append(buf[:n].copy())
            flat = np.concatenate(flat) if flat else np.zeros((0, 2), dtype=np.uint64)
            key = f"{name}_{w}x{h}"
            out[key + "_meta"] = np.array([lim, mi, len(flat)], dtype=np.uint64)
            if (w, h) in small:
                out[key + "_counts"] = np.array(counts, dtype=np.uint32)
                out[key + "_yx"] = flat.astype(np.uint32)
            else:       # hash of (count, y, x ...) per index in order
                out[key + "_fnv"] = np.array([fnv_words(counts), fnv_words((flat[:, 0] << np.uint64(32)) | flat[:, 1])], dtype=np.uint64)
        print("scan", name)
    # one-index-per-pixel methods at BASELINE's frame sizes (configs 2 and 4), hashed in C by this stub-free build: what SURVEY 8c recorded from a
    # build with a stand-in <libavutil/eval.h> (tests/golden/scan_golden.json: 7680x4320 zigzag 5107222c372da523) is reproduced without one
    sm.ref_scan_fnv1a.restype = C.c_ulonglong
    sm.ref_scan_fnv1a.argtypes = [C.c_int, st, st]
    frames = [(3840, 2160), (7680, 4320), (2160, 3840), (1920, 1080), (256, 256)]
    out["fnv1a_sizes"] = np.array(frames)
    for m, name in enumerate(SCAN_METHODS[:3]):
        for (w, h) in frames:
            out[f"{name}_{w}x{h}_fnv1a"] = np.array([sm.ref_scan_fnv1a(m, w, h)], dtype=np.uint64)
        print("scan fnv1a", name, ["%016x" % int(out[f"{name}_{w}x{h}_fnv1a"][0]) for (w, h) in frames])
    # precomputed methods: owner index per pixel and the per-index coordinate order
    for which, name, args in ((0, "radial", None), (1, "iradial", None), (0, "radial_floor", b"floor"), (1, "iradial_ceil", b"upward")):
        for (w, h) in small + [(640, 480), (1920, 1080)]:
            p = sm.ref_precomputed(which, w, h, 3, None, args).contents
            idx = np.zeros(w * h, dtype=np.uint32)
            order = []
            for i in range(p.limit):
                for k in range(p.intervals[i]):
                    y, x = p.scans[i][k][0], p.scans[i][k][1]
                    idx[y * w + x] = i
                    order.append(y * w + x)
            key = f"{name}_{w}x{h}"
            out[key + "_limit"] = np.array([p.limit], dtype=np.uint64)
            if w * h <= 33 * 20:
                out[key + "_index"] = idx
                out[key + "_order"] = np.array(order, dtype=np.uint32)
            else:
                out[key + "_fnv"] = np.array([fnv_words(idx), fnv_words(order)], dtype=np.uint64)
        print("scan", name)
    # magnitude: distinct float keys (no ties: the order is then independent of qsort's tie-breaking), and a quantised variant whose
    # GROUPS are compared as sets (scan_methods.c:263: ties keep whatever order qsort left them in)
    for ci, (w, h, q) in enumerate([(16, 9, None), (33, 20, None), (16, 9, b"64"), (33, 20, b"40")]):
        coeffs = (synth_f32(0xD5F1800 + ci, w * h * 3) * 2 - 1).astype(np.float32) / np.float32(4 * w * h)
        coeffs[:3] = 0.5
        p = sm.ref_precomputed(2, w, h, 3, coeffs.ctypes.data, q).contents
        groups, flat = [], []
        for i in range(p.limit):
            g = [int(p.scans[i][k][0]) * w + int(p.scans[i][k][1]) for k in range(p.intervals[i])]
            groups.append(len(g)); flat += g
        out[f"magnitude{ci}_shape"] = np.array([w, h, int(q) if q else 0])
        out[f"magnitude{ci}_coeffs"] = coeffs
        out[f"magnitude{ci}_group_sizes"] = np.array(groups, dtype=np.uint32)
        out[f"magnitude{ci}_order"] = np.array(flat, dtype=np.uint32)
    print("scan magnitude")
    path = os.path.join(HERE, "ref_scan.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


def compile_tu(tmp, name, text):
    src = os.path.join(tmp, name + ".c")
    so = os.path.join(tmp, name + ".so")
    with open(src, "w") as f:
        f.write(text)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-DCOEFF_PRECISION=L", "-DINTERMEDIATE_PRECISION=L", "-O2", "-fPIC", "-shared",
                           "-I" + os.path.join(REF, "include"), src, "-o", so, "-lm"])
    return C.CDLL(so)


def rnd(seed, n):
    """zero-mean values in [-1, 1) from the survey's splitmix64 stream"""
    return synth_f32(seed, n).astype(np.float64) * 2 - 1


def main():
    assert os.path.isdir(REF), REF
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        if len(sys.argv) < 2 or sys.argv[1] == "scan":
            scan_fixtures(tmp)
        if len(sys.argv) < 2 or sys.argv[1] == "motion":
            motion_fixtures(tmp)
        if len(sys.argv) < 2 or sys.argv[1] == "spec":
            spec_fixtures(tmp)
        if len(sys.argv) > 1 and sys.argv[1] in ("scan", "motion", "spec"):
            return
        sz = build_scan_zoom(tmp)
        ab = build_applybasis(tmp)
        vp = C.c_void_p
        sz.ref_pruned_idct.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        sz.ref_zoom.restype = C.c_size_t
        sz.ref_zoom.argtypes = [vp, vp, C.c_int] + [C.c_longdouble] * 6 + [C.c_size_t] * 4 + [vp]
        ab.ref_basis.argtypes = [C.c_int, C.c_longlong, C.c_longlong, C.c_ulonglong, C.c_int, vp, vp]

        def pruned(c, coords):
            h, w, ch = c.shape
            cl = np.ascontiguousarray(c, dtype=LD)
            img = np.zeros((h, w, ch), dtype=LD)
            co = np.ascontiguousarray(coords, dtype=np.uint64)   # size_t [n][2] = (y, x)
            sz.ref_pruned_idct(cl.ctypes.data, img.ctypes.data, co.ctypes.data, len(co), w, h, ch)
            return img.astype(np.float64)

        # --- (1) dense 2-D REDFT01 by scan's pruned_idct over ALL coefficients ------------------------------
        dense = [(6, 8, 1), (12, 16, 3), (16, 9, 3), (9, 16, 2), (45, 30, 3), (48, 64, 3), (60, 135, 1), (96, 80, 3)]
        out["dense_shapes"] = np.array(dense)
        for i, (h, w, ch) in enumerate(dense):
            c = rnd(0xD5F1000 + i, h * w * ch).reshape(h, w, ch)
            coords = np.array([(y, x) for y in range(h) for x in range(w)], dtype=np.uint64)
            # a non-raster visiting order: the sum must not depend on it beyond rounding
            coords = coords[np.argsort(splitmix64_stream(0xD5F1100 + i, h * w), kind="stable")]
            out[f"dense{i}_coeffs"] = c
            out[f"dense{i}_image"] = pruned(c, coords)
            print("dense", (h, w, ch))
        # the whole operator at 6x8: image of every unit coefficient
        h, w = 6, 8
        op = np.zeros((h * w, h * w))
        for j in range(h * w):
            c = np.zeros((h, w, 1)); c.flat[j] = 1
            op[:, j] = pruned(c, np.array([(j // w, j % w)], dtype=np.uint64)).ravel()
        out["operator_6x8"] = op

        # --- (2) sparse spectra at listed frame sizes, sampled output pixels --------------------------------
        sparse = [(480, 640, 3, 512, 2048), (1080, 1920, 3, 96, 4096), (2160, 3840, 3, 48, 4096)]
        out["sparse_shapes"] = np.array(sparse)
        for i, (h, w, ch, ncoef, npix) in enumerate(sparse):
            u = splitmix64_stream(0xD5F1200 + i, ncoef * 2)
            cy = (u[0::2] % np.uint64(h)).astype(np.int64); cx = (u[1::2] % np.uint64(w)).astype(np.int64)
            # a few low-frequency and edge coefficients always present
            cy[:4] = [0, 0, 1, h - 1]; cx[:4] = [0, 1, 0, w - 1]
            key = cy * w + cx
            _, first = np.unique(key, return_index=True)
            first.sort()
            cy, cx = cy[first], cx[first]
            vals = rnd(0xD5F1300 + i, len(cy) * ch).reshape(len(cy), ch)
            c = np.zeros((h, w, ch)); c[cy, cx] = vals
            img = pruned(c, np.stack([cy, cx], 1).astype(np.uint64))
            p = splitmix64_stream(0xD5F1400 + i, npix * 2)
            py = (p[0::2] % np.uint64(h)).astype(np.int64); px = (p[1::2] % np.uint64(w)).astype(np.int64)
            py[:4] = [0, 0, h - 1, h - 1]; px[:4] = [0, w - 1, 0, w - 1]
            out[f"sparse{i}_cy"], out[f"sparse{i}_cx"], out[f"sparse{i}_vals"] = cy, cx, vals
            out[f"sparse{i}_py"], out[f"sparse{i}_px"], out[f"sparse{i}_image_at"] = py, px, img[py, px]
            print("sparse", (h, w, ch), len(cy))

        # --- (3) zoom: basis + product ------------------------------------------------------------------------
        # (h, w, type, xnum, xden, ynum, yden, vx, vy)   type: 0 interpolated, 1 centered, 2 native (zoom.c:22-26)
        zoom = [(12, 16, 0, 1, 1, 1, 1, 0, 0), (12, 16, 0, 2, 1, 2, 1, 0, 0), (12, 16, 0, 4, 1, 4, 1, 0, 0), (24, 32, 0, 4, 1, 4, 1, 0, 0),
                (12, 16, 2, 2, 1, 2, 1, 0, 0), (12, 16, 2, 3, 1, 2, 1, 0, 0), (16, 16, 0, 3, 2, 3, 2, 2.5, 2.5), (12, 16, 1, 5, 2, 7, 3, 1.25, -3.5),
                (20, 12, 0, 1, 2, 3, 4, 0, 0), (12, 16, 2, 1, 2, 1, 2, 0.5, 0), (15, 9, 0, 7, 3, 2, 1, -2, 4)]
        out["zoom_cases"] = np.array(zoom, dtype=np.float64)
        for i, (h, w, typ, xn, xd, yn, yd, vx, vy) in enumerate(zoom):
            vw = int(w * xn / xd); vh = int(h * yn / yd)   # zoom.c:286-289
            c = rnd(0xD5F1500 + i, h * w * 3).reshape(h, w, 3) * (4 * w * h)
            cl = np.ascontiguousarray(c, dtype=LD)
            ic = np.zeros((vh, vw, 3), dtype=LD)
            chh = C.c_size_t(0)
            cw = sz.ref_zoom(cl.ctypes.data, ic.ctypes.data, typ, xn, xd, yn, yd, vx, vy, vw, vh, w, h, C.addressof(chh))
            out[f"zoom{i}_coeffs"] = c
            out[f"zoom{i}_out"] = ic.astype(np.float64)
            out[f"zoom{i}_ncomp"] = np.array([cw, chh.value])
            print("zoom", zoom[i], (vh, vw), (cw, chh.value))

        # --- (4) applybasis: basis tables and partial sums -----------------------------------------------------
        def table(f, N, ortho, K=None, off=0):
            K = N if K is None else K
            t = np.zeros((K, N), dtype=np.complex128)
            re = C.c_longdouble(); im = C.c_longdouble()
            for k in range(K):
                for n in range(N):
                    ab.ref_basis(f, k + off, n, N, ortho, C.addressof(re), C.addressof(im))
                    t[k, n] = complex(float(re.value), float(im.value))
            return t
        lens = [2, 3, 4, 5, 8, 15, 16, 27, 45, 48, 64, 100]
        out["basis_lens"] = np.array(lens)
        for N in lens:
            out[f"dct2_{N}"] = table(3, N, 0).real      # REDFT10: Y_k = 2 sum_n x_n dct2(k, n)
            out[f"dct3_{N}"] = table(4, N, 0).real      # REDFT01: Y_k = 2 sum_n X_n dct3(k, n)
        for N in (4, 8, 16):
            for ortho in (0, 1):
                out[f"basis_all_{N}_{ortho}"] = np.stack([table(f, N, ortho) for f in range(12)])
        for ortho in (0, 1):   # non power of two: every function but wht
            out[f"basis_all_6_{ortho}"] = np.stack([table(f, 6, ortho) for f in range(12) if FUNCS[f] != "wht"])

        class Coords(C.Union):
            _fields_ = [("a", C.c_ulonglong * 2)]

        class Offsets(C.Union):
            _fields_ = [("a", C.c_longlong * 2)]
        ab.ref_partsums.argtypes = [C.c_int, C.c_int, C.c_int, Coords, Coords, Coords, Offsets, vp, vp]

        def co(w, h, T=Coords):
            c = T(); c.a[0] = w; c.a[1] = h
            return c
        # (func, ortho, inverse, w, h, terms_w, terms_h, P_w, P_h, off_w, off_h)
        parts = [(3, 0, 0, 8, 8, 8, 8, 1, 1, 0, 0), (3, 0, 0, 8, 8, 8, 8, 8, 8, 0, 0), (4, 0, 0, 8, 8, 8, 8, 8, 8, 0, 0), (0, 0, 0, 8, 8, 4, 4, 2, 2, 1, -1),
                 (3, 1, 0, 12, 8, 6, 4, 3, 2, 2, 1), (8, 0, 0, 8, 8, 8, 8, 4, 4, 0, 0), (10, 0, 0, 8, 8, 8, 8, 2, 2, 0, 0), (11, 0, 0, 12, 8, 5, 3, 4, 2, 0, 2),
                 (4, 1, 1, 8, 8, 8, 8, 2, 2, 0, 0), (4, 0, 1, 8, 8, 8, 8, 2, 2, 1, 2), (1, 0, 1, 12, 8, 12, 8, 3, 2, 2, -1), (3, 0, 1, 12, 8, 6, 4, 1, 1, 3, 1)]
        out["parts_cases"] = np.array(parts)
        for i, (f, ortho, inv, w, h, tw, th, pw, ph, ow, oh) in enumerate(parts):
            pix = rnd(0xD5F1600 + i, h * w * 3).reshape(h, w, 3)
            pl = np.ascontiguousarray(pix, dtype=LD)
            # term counts as the loop of :410-413 runs them (after :379-380 divide N by the partial-sum block)
            K = (tw, th) if not inv else (w, h)
            N = ((w // pw), (h // ph)) if not inv else (tw // pw, th // ph)
            nterm = K[0] * K[1] * N[0] * N[1]
            o = np.zeros((nterm, 3, 2), dtype=LD)
            ab.ref_partsums(f, ortho, inv, co(w, h), co(tw, th), co(pw, ph), co(ow, oh, Offsets), pl.ctypes.data, o.ctypes.data)
            out[f"parts{i}_pix"] = pix
            out[f"parts{i}_out"] = o.astype(np.float64).reshape(K[1], K[0], N[1], N[0], 3, 2)
            print("partsums", parts[i])

        # --- (4b) the `.coeff` file in the reference's default build (applybasis/Makefile:1-2: INTERMEDIATE_PRECISION=L: 32 bytes a value) ----
        ab.ref_write_coeff.argtypes = [C.c_char_p, C.c_ulonglong, C.c_ulonglong, vp, C.c_size_t]
        cw, chh = 4, 3
        cv = rnd(0xD5F1700, cw * chh * 3 * 2).astype(LD) * LD(1e3) + LD(2) ** -60            # values that need more than a double's 53 bits
        cpath = os.path.join(tmp, "ref.coeff")
        assert ab.ref_write_coeff(cpath.encode(), cw, chh, np.ascontiguousarray(cv).ctypes.data, cw * chh) == 0
        raw = np.fromfile(cpath, dtype=np.uint8)
        # the six padding bytes of every x87 long double are whatever the stack held: zeroed here so that the fixture is reproducible
        body = raw[16:].reshape(-1, 16).copy(); body[:, 10:] = 0
        out["coeff_L_file"] = np.concatenate([raw[:16], body.ravel()])
        out["coeff_L_values"] = cv.astype(np.float64).reshape(chh, cw, 3, 2)
        print("coeff file", raw.size, "bytes")

        # --- (5) scan's `random` order: the permutation init_random draws with this libc's rand() ------------------------
        rn = build_random(tmp)
        rn.ref_init_random.restype = C.POINTER(C.c_size_t)
        rn.ref_init_random.argtypes = [C.c_size_t, C.c_size_t, C.c_char_p]
        rcases = [(16, 9, 42), (9, 16, 7), (64, 48, 123456789)]
        out["random_cases"] = np.array(rcases)
        for i, (w, h, seed) in enumerate(rcases):
            ptr = rn.ref_init_random(w, h, str(seed).encode())
            out[f"random{i}_perm"] = np.array(ptr[:w * h], dtype=np.uint64)
            print("random", (w, h, seed))

    path = os.path.join(HERE, "ref_direct.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
