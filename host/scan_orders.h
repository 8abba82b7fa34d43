/*
 * scan_orders.h -- integer scan-order generators of the reference's `scan` tool (scan/scan_methods.c) and its two
 * plaintext serialisations (scan/scan_precomputed.c:122-153), as a small plain-C library for the harnesses.
 * Coordinates are (y, x) pairs, as scan/scan_context.c:44 hands them to scan/scan.c:431.
 */
#ifndef SCAN_ORDERS_H
#define SCAN_ORDERS_H
#include <stddef.h>
#include <stdio.h>

enum scan_order_method {
	SCAN_HORIZONTAL, SCAN_VERTICAL, SCAN_ZIGZAG,           /* one coordinate per index (scan_methods.c:59-115) */
	SCAN_ROW, SCAN_COLUMN, SCAN_DIAGONAL, SCAN_MIRROR,      /* several per index (:146-187) */
	SCAN_BOX, SCAN_IBOX,                                    /* :122-144, including their out-of-range / duplicate quirks */
	SCAN_RADIAL, SCAN_IRADIAL,                              /* :298-331, default rounding (rint) */
	SCAN_METHOD_COUNT
};

/* shortest method name with the given prefix (scan_methods.c:581-591); -1 if none */
int scan_order_find_prefix(const char *prefix);
const char *scan_order_name(int method);

size_t scan_order_limit(int method, size_t w, size_t h);                 /* number of scan indices */
size_t scan_order_max_interval(int method, size_t w, size_t h);          /* most coordinates one index can yield */
/* coordinates of scan index i into yx[][2] (room for scan_order_max_interval entries); returns how many */
size_t scan_order_coords(int method, size_t w, size_t h, size_t i, size_t (*yx)[2]);

/* The `file` method (scan_methods.c:393-410 + scan_precomputed.c:24-120): reads either plaintext serialisation -- coordinate
 * lists ("x,y x,y ..." per scan index, one index per line; detected by a ',' or an empty first line) or an index grid (one scan
 * index per pixel, row by row; blank lines skipped) -- and rejects coordinates outside w x h, as init_file does.
 * On success returns 0 and fills *out (free with scan_order_list_free): coordinates of index i are yx[offset[i] .. offset[i+1]). */
struct scan_order_list { size_t limit, max_interval, total; size_t *offset; size_t (*yx)[2]; };
int scan_order_read_file(FILE *f, size_t w, size_t h, struct scan_order_list *out);
void scan_order_list_free(struct scan_order_list *l);

/* The `random` method (scan_methods.c:210-228 init_random + :117-120 scan_ordered): a permutation of the pixels drawn with libc's
 * srand(seed) / rand() exactly as the tool draws it (Fisher-Yates from the top that stops above index 1), one coordinate per scan
 * index.  Reproducible for a given libc and seed; the tool's default seed is time(NULL).  Returns 0 and fills *out like
 * scan_order_read_file. */
int scan_order_random(size_t w, size_t h, unsigned int seed, struct scan_order_list *out);

/* scan_precomputed.c:122-153.  Return 0 on success. */
int scan_order_serialize_coordinate(int method, size_t w, size_t h, FILE *f);
int scan_order_serialize_index(int method, size_t w, size_t h, FILE *f);
#endif
