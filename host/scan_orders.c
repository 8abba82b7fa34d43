/* scan_orders.c -- see scan_orders.h.  Each generator cites the reference lines it follows. */
#include "scan_orders.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const char *const names[SCAN_METHOD_COUNT] = {
	"horizontal", "vertical", "zigzag", "row", "column", "diagonal", "mirror", "box", "ibox", "radial", "iradial"};

const char *scan_order_name(int m) { return m >= 0 && m < SCAN_METHOD_COUNT ? names[m] : NULL; }

int scan_order_find_prefix(const char *prefix)
{
	size_t len = strlen(prefix), best = (size_t)-1;
	int found = -1;
	for (int m = 0; m < SCAN_METHOD_COUNT; m++)
		if (!strncmp(names[m], prefix, len) && strlen(names[m]) < best) { best = strlen(names[m]); found = m; }
	return found;
}

static size_t umin(size_t a, size_t b) { return a < b ? a : b; }
static size_t umax(size_t a, size_t b) { return a > b ? a : b; }

/* radial / iradial bucket every pixel by its rounded distance (scan_methods.c:298-331); the buckets are kept
 * in raster order, which is the order scan_precomputed_add_coord appends them in */
static size_t radial_index(int method, size_t w, size_t h, size_t x, size_t y)
{
	if (method == SCAN_RADIAL) return (size_t)rint(hypot((double)x, (double)y));
	size_t limit = (size_t)rint(hypot((double)(w - 1), (double)(h - 1))) + 1;
	return limit - (size_t)rint(hypot((double)(w - x - 1), (double)(h - y - 1))) - 1;
}

size_t scan_order_limit(int method, size_t w, size_t h)
{
	switch (method) {
	case SCAN_ROW: return h;                                       /* limit_height */
	case SCAN_COLUMN: return w;                                    /* limit_width */
	case SCAN_DIAGONAL: return w + h - 1;                          /* limit_sum */
	case SCAN_MIRROR: case SCAN_BOX: return umax(w, h);            /* limit_max */
	case SCAN_IBOX: return umin(w, h);                             /* limit_min */
	case SCAN_RADIAL: case SCAN_IRADIAL: {
		size_t lim = 0;
		for (size_t y = 0; y < h; y++) for (size_t x = 0; x < w; x++) lim = umax(lim, radial_index(method, w, h, x, y) + 1);
		return lim;
	}
	default: return w * h;
	}
}

size_t scan_order_max_interval(int method, size_t w, size_t h)
{
	switch (method) {
	case SCAN_ROW: return w;
	case SCAN_COLUMN: return h;
	case SCAN_DIAGONAL: return umin(w, h);                         /* limit_min */
	case SCAN_MIRROR: return umin(w, h) * 2 - 1;                   /* limit_mirror */
	case SCAN_BOX: case SCAN_IBOX: return w + h - 1;               /* limit_sum (scan_methods.c:23,496,502).  ibox index 0 emits its corner twice = w + h
	                                                                * coordinates: callers allocate max_interval + 1 entries, as scan.c:346 does */
	case SCAN_RADIAL: case SCAN_IRADIAL: {
		size_t lim = scan_order_limit(method, w, h), best = 0;
		size_t *cnt = calloc(lim, sizeof *cnt);
		for (size_t y = 0; y < h; y++) for (size_t x = 0; x < w; x++) cnt[radial_index(method, w, h, x, y)]++;
		for (size_t i = 0; i < lim; i++) best = umax(best, cnt[i]);
		free(cnt);
		return best;
	}
	default: return 1;
	}
}

static size_t tri(size_t d) { return d * (d + 1) / 2; }
static size_t tri_floor(size_t i) { return (size_t)(sqrt((double)(i * 2) + 0.25) - 0.5); }   /* scan_methods.c:69-71 */

size_t scan_order_coords(int method, size_t w, size_t h, size_t i, size_t (*yx)[2])
{
	size_t n = 0;
	switch (method) {
	case SCAN_HORIZONTAL: yx[0][0] = i / w; yx[0][1] = i % w; return 1;                 /* :59-62 */
	case SCAN_VERTICAL: yx[0][0] = i % h; yx[0][1] = i / h; return 1;                   /* :64-67 */
	case SCAN_ZIGZAG: {                                                                  /* :77-115 */
		const size_t m = umin(w, h), head = tri(m), area = w * h;
		if (i < head) {
			size_t d = tri_floor(i), r = i - tri(d);
			if (!(d & 1)) r = d - r;
			yx[0][0] = r; yx[0][1] = d - r;
		} else if (area - i <= head) {
			size_t j = area - i - 1, d = tri_floor(j), r = j - tri(d);
			if (!(((w + h - 1) - d - 1) & 1)) r = d - r;
			yx[0][0] = (h - 1) - r; yx[0][1] = (w - 1) - (d - r);
		} else {
			size_t band = (i - head) / m, r = m - (i - (band * m + head));
			if (!((band + m) & 1)) r = m - r + 1;
			if (w < h) { r = m - r + 1; yx[0][0] = band + r; yx[0][1] = w - r; }
			else { yx[0][0] = h - r; yx[0][1] = band + r; }
		}
		return 1;
	}
	case SCAN_ROW: for (size_t x = 0; x < w; x++) { yx[x][0] = i; yx[x][1] = x; } return w;           /* :146-151 */
	case SCAN_COLUMN: for (size_t y = 0; y < h; y++) { yx[y][0] = y; yx[y][1] = i; } return h;         /* :153-158 */
	case SCAN_DIAGONAL: {                                                                               /* :160-165 */
		size_t y = i < h ? i : h - 1, x = i - y;
		for (; x < w; x++) { yx[n][0] = y; yx[n][1] = x; n++; if (!y) break; y--; }
		return n;
	}
	case SCAN_MIRROR:                                                                                   /* :167-187 */
		if (i > 0) {
			if (i < w) for (size_t x = umin(h, w - i); x > 0; x--, n++) { yx[n][0] = x - 1; yx[n][1] = x + i - 1; }
			if (i < h) for (size_t y = umin(w, h - i); y > 0; y--, n++) { yx[n][0] = y + i - 1; yx[n][1] = y - 1; }
		} else for (size_t d = 0; d < umin(w, h); d++, n++) yx[d][0] = yx[d][1] = d;
		return n;
	case SCAN_BOX: {                                                                                    /* :122-133 (x = i is NOT clamped on the first leg) */
		size_t ymax = i < h ? i : h - 1, xmax = i < w ? i : w - 1;
		for (size_t y = 0; y < ymax; y++, n++) { yx[n][0] = y; yx[n][1] = i; }
		for (size_t x = 0; x < xmax + 1; x++, n++) { yx[n][0] = ymax; yx[n][1] = x; }
		return n;
	}
	case SCAN_IBOX:                                                                                     /* :135-144 (corner (i,i) emitted twice) */
		for (size_t x = i; x < w; x++, n++) { yx[n][0] = i; yx[n][1] = x; }
		for (size_t y = i; y < h; y++, n++) { yx[n][0] = y; yx[n][1] = i; }
		return n;
	case SCAN_RADIAL: case SCAN_IRADIAL:
		for (size_t y = 0; y < h; y++)
			for (size_t x = 0; x < w; x++)
				if (radial_index(method, w, h, x, y) == i) { yx[n][0] = y; yx[n][1] = x; n++; }
		return n;
	}
	return 0;
}

int scan_order_serialize_coordinate(int method, size_t w, size_t h, FILE *f)
{
	size_t lim = scan_order_limit(method, w, h);
	size_t (*yx)[2] = malloc(sizeof(*yx) * (scan_order_max_interval(method, w, h) + 1));
	for (size_t i = 0; i < lim; i++) {
		size_t n = scan_order_coords(method, w, h, i, yx);
		for (size_t j = 0; j < n; j++) if (fprintf(f, "%zu,%zu ", yx[j][1], yx[j][0]) <= 0) { free(yx); return 1; }   /* x,y (scan_precomputed.c:125) */
		if (fprintf(f, "\n") <= 0) { free(yx); return 1; }
	}
	free(yx);
	return 0;
}

int scan_order_serialize_index(int method, size_t w, size_t h, FILE *f)
{
	size_t lim = scan_order_limit(method, w, h);
	int pad = (int)(log10f((float)lim) + 1);                                                            /* scan_precomputed.c:135 */
	size_t (*yx)[2] = malloc(sizeof(*yx) * (scan_order_max_interval(method, w, h) + 1));
	size_t ow = 0, oh = 0;                                                                              /* :10-22: dimensions from the coordinates */
	for (size_t i = 0; i < lim; i++) {
		size_t n = scan_order_coords(method, w, h, i, yx);
		for (size_t j = 0; j < n; j++) { oh = umax(oh, yx[j][0]); ow = umax(ow, yx[j][1]); }
	}
	ow++; oh++;
	size_t *index = calloc(ow * oh, sizeof *index);
	for (size_t i = 0; i < lim; i++) {
		size_t n = scan_order_coords(method, w, h, i, yx);
		for (size_t j = 0; j < n; j++) index[yx[j][0] * ow + yx[j][1]] = i;
	}
	int err = 0;
	for (size_t y = 0; y < oh && !err; y++) {
		for (size_t x = 0; x < ow && !err; x++) err = fprintf(f, "%*zu ", pad, index[y * ow + x]) <= 0;
		if (!err) err = fprintf(f, "\n") <= 0;
	}
	free(index); free(yx);
	return err;
}

/* ---- the `file` method: scan_precomputed.c:24-120 ---- */
struct pre { size_t limit, *intervals; size_t (**scans)[2]; };

static int pre_add(struct pre *p, size_t index, size_t x, size_t y)                   /* scan_precomputed.c:24-49 */
{
	if (index >= p->limit) {
		size_t *iv = realloc(p->intervals, sizeof(*iv) * (index + 1));
		if (!iv) return 0;
		p->intervals = iv;
		size_t (**sc)[2] = realloc(p->scans, sizeof(*sc) * (index + 1));
		if (!sc) return 0;
		p->scans = sc;
		for (size_t i = p->limit; i <= index; i++) { p->intervals[i] = 0; p->scans[i] = NULL; }
		p->limit = index + 1;
	}
	size_t (*s)[2] = realloc(p->scans[index], sizeof(*s) * (p->intervals[index] + 1));
	if (!s) return 0;
	p->scans[index] = s;
	s[p->intervals[index]][0] = y; s[p->intervals[index]][1] = x;
	p->intervals[index]++;
	return 1;
}
static void pre_free(struct pre *p)
{
	for (size_t i = 0; i < p->limit; i++) free(p->scans[i]);
	free(p->scans); free(p->intervals);
	memset(p, 0, sizeof *p);
}

int scan_order_read_file(FILE *f, size_t w, size_t h, struct scan_order_list *out)
{
	memset(out, 0, sizeof *out);
	struct pre p = {0, NULL, NULL};
	char *line = NULL;
	size_t cap = 0;
	int ok = 0;
	if (getline(&line, &cap, f) <= 0) { free(line); return 1; }
	const int coordinate = strchr(line, ',') || *line == '\n';                         /* scan_precomputed.c:108 */
	size_t row = 0;
	do {
		if (!coordinate && *line == '\n') continue;                                    /* :84-85: blank lines of an index grid */
		char *string = line, *token;
		size_t col = 0;
		while ((token = strsep(&string, " ")) && *token != '\n') {
			if (!*token) continue;
			size_t a, b;
			if (coordinate) { if (sscanf(token, "%zu,%zu", &a, &b) != 2 || !pre_add(&p, row, a, b)) goto done; }           /* x,y of index `row` */
			else { if (sscanf(token, "%zu", &a) != 1 || !pre_add(&p, a, col, row)) goto done; col++; }                     /* index of pixel (col, row) */
		}
		row++;
	} while (getline(&line, &cap, f) > 0);
	if (!feof(f) || !p.limit) goto done;
	for (size_t i = 0; i < p.limit; i++)                                               /* scan_methods.c:401-407 */
		for (size_t j = 0; j < p.intervals[i]; j++)
			if (p.scans[i][j][1] >= w || p.scans[i][j][0] >= h) goto done;
	out->limit = p.limit;
	out->offset = malloc(sizeof(size_t) * (p.limit + 1));
	for (size_t i = 0; i < p.limit; i++) { out->offset[i] = out->total; out->total += p.intervals[i]; out->max_interval = umax(out->max_interval, p.intervals[i]); }
	out->offset[p.limit] = out->total;
	out->yx = malloc(sizeof(*out->yx) * (out->total + 1));
	for (size_t i = 0; i < p.limit; i++) memcpy(out->yx + out->offset[i], p.scans[i], sizeof(*out->yx) * p.intervals[i]);
	ok = 1;
done:
	free(line);
	pre_free(&p);
	return ok ? 0 : 1;
}

int scan_order_random(size_t w, size_t h, unsigned int seed, struct scan_order_list *out)
{
	memset(out, 0, sizeof *out);
	const size_t len = w * h;
	if (!len) return 1;
	size_t *ctx = malloc(sizeof(size_t) * len);
	out->offset = malloc(sizeof(size_t) * (len + 1));
	out->yx = malloc(sizeof(*out->yx) * (len + 1));
	if (!ctx || !out->offset || !out->yx) { free(ctx); scan_order_list_free(out); return 1; }
	srand(seed);                                                                       /* scan_methods.c:217 */
	for (size_t i = 0; i < len; i++) ctx[i] = i;
	for (size_t i = len - 1; i > 1; i--) {                                             /* :220-225: the loop ends above index 1 */
		const size_t j = (size_t)rand() % (i + 1);
		const size_t tmp = ctx[j];
		ctx[j] = ctx[i];
		ctx[i] = tmp;
	}
	for (size_t i = 0; i < len; i++) { out->offset[i] = i; out->yx[i][0] = ctx[i] / w; out->yx[i][1] = ctx[i] % w; }   /* :117-120 */
	out->offset[len] = len;
	out->limit = out->total = len;
	out->max_interval = 1;
	free(ctx);
	return 0;
}

void scan_order_list_free(struct scan_order_list *l)
{
	free(l->offset); free(l->yx);
	memset(l, 0, sizeof *l);
}

