/* expr_eval.c -- see expr_eval.h: a recursive-descent parser into a small tree, evaluated per call. */
#include "expr_eval.h"
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum op {
	O_CONST, O_VAR, O_ADD, O_SUB, O_MUL, O_DIV, O_POW, O_NEG, O_SEQ,
	/* functions of one argument */
	F_ABS, F_ACOS, F_ASIN, F_ATAN, F_CEIL, F_COS, F_COSH, F_EXP, F_FLOOR, F_GAUSS, F_ISINF, F_ISNAN, F_LD, F_LOG, F_NOT, F_ROUND, F_SGN, F_SIN, F_SINH,
	F_SQRT, F_SQUISH, F_TAN, F_TANH, F_TRUNC,
	/* two */
	F_ATAN2, F_BITAND, F_BITOR, F_EQ, F_GCD, F_GT, F_GTE, F_HYPOT, F_LT, F_LTE, F_MAX, F_MIN, F_MOD, F_FPOW, F_ST, F_WHILE,
	/* two or three */
	F_IF, F_IFNOT,
	/* three */
	F_BETWEEN, F_CLIP, F_LERP
};
struct expr { enum op op; double value; int var; struct expr *a[3]; double *store; /* the ten st() / ld() variables, owned by the root */ };

static const struct { const char *name; enum op op; int min, max; } funcs[] = {
	{"abs", F_ABS, 1, 1}, {"acos", F_ACOS, 1, 1}, {"asin", F_ASIN, 1, 1}, {"atan", F_ATAN, 1, 1}, {"ceil", F_CEIL, 1, 1}, {"cos", F_COS, 1, 1}, {"cosh", F_COSH, 1, 1},
	{"exp", F_EXP, 1, 1}, {"floor", F_FLOOR, 1, 1}, {"gauss", F_GAUSS, 1, 1}, {"isinf", F_ISINF, 1, 1}, {"isnan", F_ISNAN, 1, 1}, {"ld", F_LD, 1, 1}, {"log", F_LOG, 1, 1},
	{"not", F_NOT, 1, 1}, {"round", F_ROUND, 1, 1}, {"sgn", F_SGN, 1, 1}, {"sin", F_SIN, 1, 1}, {"sinh", F_SINH, 1, 1}, {"sqrt", F_SQRT, 1, 1}, {"squish", F_SQUISH, 1, 1},
	{"tan", F_TAN, 1, 1}, {"tanh", F_TANH, 1, 1}, {"trunc", F_TRUNC, 1, 1},
	{"atan2", F_ATAN2, 2, 2}, {"bitand", F_BITAND, 2, 2}, {"bitor", F_BITOR, 2, 2}, {"eq", F_EQ, 2, 2}, {"gcd", F_GCD, 2, 2}, {"gt", F_GT, 2, 2}, {"gte", F_GTE, 2, 2},
	{"hypot", F_HYPOT, 2, 2}, {"lt", F_LT, 2, 2}, {"lte", F_LTE, 2, 2}, {"max", F_MAX, 2, 2}, {"min", F_MIN, 2, 2}, {"mod", F_MOD, 2, 2}, {"pow", F_FPOW, 2, 2},
	{"st", F_ST, 2, 2}, {"while", F_WHILE, 2, 2}, {"if", F_IF, 2, 3}, {"ifnot", F_IFNOT, 2, 3}, {"between", F_BETWEEN, 3, 3}, {"clip", F_CLIP, 3, 3}, {"lerp", F_LERP, 3, 3},
};

struct parser { const char *s; const char *const *names; int err; };

static struct expr *node(enum op op, struct expr *a, struct expr *b, struct expr *c)
{
	struct expr *e = calloc(1, sizeof *e);
	if (e) { e->op = op; e->a[0] = a; e->a[1] = b; e->a[2] = c; }
	return e;
}
static void skip(struct parser *p) { while (isspace((unsigned char)*p->s)) p->s++; }
static struct expr *parse_seq(struct parser *p);

/* a number with the postfixes of the documented syntax: SI prefixes (optionally followed by `i`: powers of 1024), `B` (times 8), `dB` */
static double number(struct parser *p)
{
	char *end;
	double d = strtod(p->s, &end);
	if (end == p->s) { p->err = 1; return 0; }
	p->s = end;
	if (p->s[0] == 'd' && p->s[1] == 'B') { p->s += 2; return pow(10.0, d / 20.0); }
	static const char pre[] = "yzafpnumcdhkKMGTPEZY";
	static const int exp10[] = {-24, -21, -18, -15, -12, -9, -6, -3, -2, -1, 2, 3, 3, 6, 9, 12, 15, 18, 21, 24};
	const char *q = *p->s ? strchr(pre, *p->s) : NULL;
	if (q && !isalpha((unsigned char)p->s[1] == 'i' ? (unsigned char)p->s[2] : (unsigned char)p->s[1])) {
		const int e10 = exp10[q - pre];
		if (p->s[1] == 'i') { if (e10 >= 3 && e10 % 3 == 0) { d *= pow(2.0, e10 / 3 * 10); p->s += 2; } else p->err = 1; }
		else { d *= pow(10.0, e10); p->s++; }
	}
	if (*p->s == 'B') { d *= 8; p->s++; }
	return d;
}

static struct expr *parse_primary(struct parser *p)
{
	skip(p);
	if (*p->s == '(') {
		p->s++;
		struct expr *e = parse_seq(p);
		skip(p);
		if (*p->s != ')') { p->err = 1; expr_free(e); return NULL; }
		p->s++;
		return e;
	}
	if (isdigit((unsigned char)*p->s) || *p->s == '.') {
		struct expr *e = node(O_CONST, NULL, NULL, NULL);
		if (e) e->value = number(p);
		return e;
	}
	if (isalpha((unsigned char)*p->s) || *p->s == '_') {
		const char *b = p->s;
		while (isalnum((unsigned char)*p->s) || *p->s == '_') p->s++;
		const size_t n = (size_t)(p->s - b);
		skip(p);
		if (*p->s == '(') {                                   /* function call */
			int f = -1;
			for (size_t i = 0; i < sizeof funcs / sizeof funcs[0]; i++) if (strlen(funcs[i].name) == n && !strncmp(funcs[i].name, b, n)) f = (int)i;
			if (f < 0) { p->err = 1; return NULL; }
			p->s++;
			struct expr *arg[3] = {NULL, NULL, NULL};
			int na = 0;
			for (;;) {
				if (na == 3) { p->err = 1; break; }
				arg[na] = parse_seq(p);
				if (!arg[na]) { p->err = 1; break; }
				na++;
				skip(p);
				if (*p->s == ',') { p->s++; continue; }
				break;
			}
			if (!p->err && *p->s == ')' && na >= funcs[f].min && na <= funcs[f].max) { p->s++; return node(funcs[f].op, arg[0], arg[1], arg[2]); }
			p->err = 1;
			for (int i = 0; i < 3; i++) expr_free(arg[i]);
			return NULL;
		}
		for (int v = 0; p->names && p->names[v]; v++)
			if (strlen(p->names[v]) == n && !strncmp(p->names[v], b, n)) { struct expr *e = node(O_VAR, NULL, NULL, NULL); if (e) e->var = v; return e; }
		static const struct { const char *name; double v; } consts[] = {{"PI", 3.14159265358979323846}, {"E", 2.7182818284590452354}, {"PHI", 1.6180339887498948482}};
		for (size_t i = 0; i < 3; i++)
			if (strlen(consts[i].name) == n && !strncmp(consts[i].name, b, n)) { struct expr *e = node(O_CONST, NULL, NULL, NULL); if (e) e->value = consts[i].v; return e; }
		p->err = 1;
		return NULL;
	}
	p->err = 1;
	return NULL;
}
/* an optional sign in front of a primary; the sign of the BASE applies to the whole power (-2^2 = -4), the exponent's to the exponent */
static struct expr *parse_signed(struct parser *p, int *neg)
{
	skip(p);
	*neg = 0;
	if (*p->s == '+') p->s++; else if (*p->s == '-') { *neg = 1; p->s++; }
	return parse_primary(p);
}
static struct expr *parse_factor(struct parser *p)
{
	int neg0, neg;
	struct expr *e = parse_signed(p, &neg0);
	for (skip(p); e && *p->s == '^'; skip(p)) {
		p->s++;
		struct expr *x = parse_signed(p, &neg);
		if (!x) { expr_free(e); return NULL; }
		e = node(O_POW, e, neg ? node(O_NEG, x, NULL, NULL) : x, NULL);
	}
	return e && neg0 ? node(O_NEG, e, NULL, NULL) : e;
}
static struct expr *parse_term(struct parser *p)
{
	struct expr *e = parse_factor(p);
	for (skip(p); e && (*p->s == '*' || *p->s == '/'); skip(p)) {
		const char c = *p->s++;
		struct expr *x = parse_factor(p);
		if (!x) { expr_free(e); return NULL; }
		e = node(c == '*' ? O_MUL : O_DIV, e, x, NULL);
	}
	return e;
}
static struct expr *parse_sum(struct parser *p)
{
	struct expr *e = parse_term(p);
	for (skip(p); e && (*p->s == '+' || *p->s == '-'); skip(p)) {
		const char c = *p->s++;
		struct expr *x = parse_term(p);
		if (!x) { expr_free(e); return NULL; }
		e = node(c == '+' ? O_ADD : O_SUB, e, x, NULL);
	}
	return e;
}
static struct expr *parse_seq(struct parser *p)
{
	struct expr *e = parse_sum(p);
	for (skip(p); e && *p->s == ';'; skip(p)) {
		p->s++;
		skip(p);
		if (!*p->s || *p->s == ')') break;                   /* a trailing `;` */
		struct expr *x = parse_sum(p);
		if (!x) { expr_free(e); return NULL; }
		e = node(O_SEQ, e, x, NULL);
	}
	return e;
}

struct expr *expr_parse(const char *s, const char *const *names)
{
	struct parser p = {s, names, 0};
	struct expr *e = parse_seq(&p);
	skip(&p);
	if (!e || p.err || *p.s) { expr_free(e); return NULL; }
	e->store = calloc(10, sizeof(double));
	if (!e->store) { expr_free(e); return NULL; }
	return e;
}
void expr_free(struct expr *e)
{
	if (!e) return;
	for (int i = 0; i < 3; i++) expr_free(e->a[i]);
	free(e->store);
	free(e);
}

static double ev(const struct expr *e, const double *v, double *st)
{
	switch (e->op) {
	case O_CONST: return e->value;
	case O_VAR: return v[e->var];
	case O_NEG: return -ev(e->a[0], v, st);
	case O_SEQ: ev(e->a[0], v, st); return ev(e->a[1], v, st);
	case F_IF: case F_IFNOT: {
		const double c = ev(e->a[0], v, st);
		const int take = e->op == F_IF ? c != 0 : c == 0;
		return take ? ev(e->a[1], v, st) : e->a[2] ? ev(e->a[2], v, st) : 0.0;
	}
	case F_WHILE: { double r = NAN; while (ev(e->a[0], v, st) != 0) r = ev(e->a[1], v, st); return r; }
	case F_ST: { const double i = ev(e->a[0], v, st), x = ev(e->a[1], v, st); const int k = i >= 0 && i < 10 ? (int)i : 0; return st[k] = x; }
	case F_LD: { const double i = ev(e->a[0], v, st); return st[i >= 0 && i < 10 ? (int)i : 0]; }
	default: break;
	}
	const double a = ev(e->a[0], v, st), b = e->a[1] ? ev(e->a[1], v, st) : 0.0, c = e->a[2] ? ev(e->a[2], v, st) : 0.0;
	switch (e->op) {
	case O_ADD: return a + b;
	case O_SUB: return a - b;
	case O_MUL: return a * b;
	case O_DIV: return a / b;
	case O_POW: case F_FPOW: return pow(a, b);
	case F_ABS: return fabs(a);
	case F_ACOS: return acos(a);
	case F_ASIN: return asin(a);
	case F_ATAN: return atan(a);
	case F_CEIL: return ceil(a);
	case F_COS: return cos(a);
	case F_COSH: return cosh(a);
	case F_EXP: return exp(a);
	case F_FLOOR: return floor(a);
	case F_GAUSS: return exp(-a * a / 2) / sqrt(2 * 3.14159265358979323846);
	case F_ISINF: return isinf(a) ? 1.0 : 0.0;
	case F_ISNAN: return isnan(a) ? 1.0 : 0.0;
	case F_LOG: return log(a);
	case F_NOT: return a == 0 ? 1.0 : 0.0;
	case F_ROUND: return round(a);
	case F_SGN: return (a > 0) - (a < 0);
	case F_SIN: return sin(a);
	case F_SINH: return sinh(a);
	case F_SQRT: return sqrt(a);
	case F_SQUISH: return 1 / (1 + exp(4 * a));
	case F_TAN: return tan(a);
	case F_TANH: return tanh(a);
	case F_TRUNC: return trunc(a);
	case F_ATAN2: return atan2(a, b);
	case F_BITAND: return isnan(a) || isnan(b) ? NAN : (double)((long)a & (long)b);
	case F_BITOR: return isnan(a) || isnan(b) ? NAN : (double)((long)a | (long)b);
	case F_EQ: return a == b ? 1.0 : 0.0;
	case F_GT: return a > b ? 1.0 : 0.0;
	case F_GTE: return a >= b ? 1.0 : 0.0;
	case F_LT: return a < b ? 1.0 : 0.0;
	case F_LTE: return a <= b ? 1.0 : 0.0;
	case F_GCD: { long x = labs((long)a), y = labs((long)b); while (y) { const long t = x % y; x = y; y = t; } return (double)x; }
	case F_HYPOT: return hypot(a, b);
	case F_MAX: return a > b ? a : b;
	case F_MIN: return a < b ? a : b;
	case F_MOD: return a - floor(a / b) * b;
	case F_BETWEEN: return a >= b && a <= c ? 1.0 : 0.0;
	case F_CLIP: return b > c ? NAN : a < b ? b : a > c ? c : a;
	case F_LERP: return a + (b - a) * c;
	default: return NAN;
	}
}
double expr_eval(const struct expr *e, const double *vars) { return ev(e, vars, e->store); }
