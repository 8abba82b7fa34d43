"""TEST INFRASTRUCTURE ONLY: pure-Python restatement (small cases) of the multi-coordinate scan generators of
scan/scan_methods.c that oracle/scan_oracle.c does not cover: mirror (:167-187), box (:122-133), ibox (:135-144),
radial / iradial (:298-331, default rounding rint), with limit/interval rules from :19-47 and the method table :453-567.
Returns, per scan index, the list of (y, x) pairs."""
import math


def _rint(v):
    return int(round(v)) if abs(v - math.floor(v) - 0.5) > 1e-12 else int(2 * round(v / 2.0))   # ties to even, like rint()


def orders(method, w, h):
    out = []
    if method == "mirror":                     # limit = max(w, h)
        for i in range(max(w, h)):
            c = []
            if i > 0:
                if i < w:
                    for x in range(min(h, w - i), 0, -1):
                        c.append((x - 1, x + i - 1))
                if i < h:
                    for y in range(min(w, h - i), 0, -1):
                        c.append((y + i - 1, y - 1))
            else:
                c = [(d, d) for d in range(min(w, h))]
            out.append(c)
    elif method == "box":                      # limit = max(w, h); first leg keeps x = i even when i >= w
        for i in range(max(w, h)):
            ymax = i if i < h else h - 1
            xmax = i if i < w else w - 1
            out.append([(y, i) for y in range(ymax)] + [(ymax, x) for x in range(xmax + 1)])
    elif method == "ibox":                     # limit = min(w, h); the corner appears in both legs
        for i in range(min(w, h)):
            out.append([(i, x) for x in range(i, w)] + [(y, i) for y in range(i, h)])
    elif method in ("radial", "iradial"):
        buckets = {}
        limit = _rint(math.hypot(w - 1, h - 1)) + 1
        for y in range(h):
            for x in range(w):
                idx = _rint(math.hypot(x, y)) if method == "radial" else limit - _rint(math.hypot(w - x - 1, h - y - 1)) - 1
                buckets.setdefault(idx, []).append((y, x))
        out = [buckets.get(i, []) for i in range(max(buckets) + 1)]
    else:
        raise ValueError(method)
    return out
