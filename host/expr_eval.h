/*
 * expr_eval.h -- arithmetic expressions of the kind the reference's tools hand to libavutil's av_expr_parse / av_expr_eval
 * (scan's `evalxy` / `evali` methods: scan/scan_methods.c:186-201,333-391; scan/README.md:105-109 `bitand(x,y)`,
 * `mod(i,height); floor(i/height)`).
 *
 * libavutil is a third-party dependency that is ABSENT from /root/reference and from this image (no pinned version: whatever
 * `pkg-config libavutil` finds), so this is a restatement of its PUBLISHED expression language (ffmpeg-utils(1), "Expression
 * Evaluation"), own code: numbers with SI / binary / dB postfixes; + - * / ^, unary sign (applied to the whole power, as libavutil's
 * parser does), parentheses, `;` sequences; the constants PI, E, PHI; named variables; and the functions listed in expr_eval.c.
 * PARITY UNPINNED for the expression language itself (nothing of it can run here); what scan does with the VALUES -- rint, the
 * rejection of NaN / infinities / negatives, `% width`, the order of coordinates within an index -- follows scan_methods.c line by line.
 * Not implemented (return a parse error): random, print, time, taylor, root.
 */
#ifndef HOST_EXPR_EVAL_H
#define HOST_EXPR_EVAL_H
struct expr;
/* names: NULL-terminated list of variable names; the values are passed to expr_eval in the same order.  NULL on a syntax error. */
struct expr *expr_parse(const char *s, const char *const *names);
double expr_eval(const struct expr *e, const double *vars);
void expr_free(struct expr *e);
#endif
