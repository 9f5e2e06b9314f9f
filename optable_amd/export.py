"""On-disk formats of traced rays, written from arrays.

`OpticalTable.export_rays_csv` (optical_table.py:447-500) writes one CSV row per segment with Mathematica-style
text fields — origin, transform_matrix (the rotation taking +x onto the direction, base.py:81-104), intensity,
length, qo, n — and `Monitor.export_rays_npz` (monitor.py:255-269) five arrays of a monitor's hits.  The reference
builds both from a `Ray` object per segment; here the columns of a `SegmentBatch` (or of a list of rays) are turned
into the same text / arrays directly, so a million-segment history does not have to become a million objects.
"""
import csv

import numpy as np


def math_str(s: str) -> str:
    """base.py:248-251."""
    if s == "None":
        return "None"
    return s.replace("[", "{").replace("]", "}").replace("e", "*10^").replace("j", "I")


def vector_to_R_batch(direction):
    """`Vector._vector_to_R` (base.py:81-104) for N directions at once.  Returns (R [N,3,3], aligned [N] bool,
    opposite [N] bool): the two colinear special cases are reported so that the writer can print them the way the
    reference's literals print (np.eye -> floats, the hard-coded 180-degree matrix -> ints)."""
    d = np.asarray(direction, dtype=float).reshape(-1, 3)
    v = d / np.linalg.norm(d, axis=1, keepdims=True)
    tol = 1e-8 + 1e-5 * np.abs(np.array([1.0, 0.0, 0.0]))  # np.allclose(v, [+-1, 0, 0]): atol + rtol * |b|
    aligned = np.all(np.abs(v - [1.0, 0.0, 0.0]) <= tol, axis=1)
    opposite = np.all(np.abs(v - [-1.0, 0.0, 0.0]) <= tol, axis=1)
    k = np.stack([np.zeros(len(v)), -v[:, 2], v[:, 1]], axis=1)  # cross([1, 0, 0], v)
    norm = np.linalg.norm(k, axis=1, keepdims=True)
    k = np.divide(k, norm, out=np.zeros_like(k), where=norm > 0)
    c = v[:, 0]
    s = np.sin(np.arccos(np.clip(c, -1.0, 1.0)))
    K = np.zeros((len(v), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    R = c[:, None, None] * np.eye(3) + (1 - c)[:, None, None] * (k[:, :, None] * k[:, None, :]) + s[:, None, None] * K
    R[aligned] = np.eye(3)
    R[opposite] = np.array([[-1.0, 0, 0], [0, -1.0, 0], [0, 0, 1.0]])
    return R, aligned, opposite


_OPPOSITE_TEXT = math_str(str([[-1, 0, 0], [0, -1, 0], [0, 0, 1]]))  # the reference returns an int array there

HEADER = ("origin", "transform_matrix", "intensity", "length", "qo", "n")


def rays_csv_rows(cols, has_q):
    """Rows of `gather_rays_csv` (optical_table.py:447-467) from columns: `cols` holds ox oy oz dx dy dz length
    intensity q_re q_im n (numpy arrays of one length, +inf length = None), `has_q` a bool per row."""
    n = len(cols["ox"])
    R, _, opposite = vector_to_R_batch(np.stack([cols["dx"], cols["dy"], cols["dz"]], axis=1)) if n else (np.zeros((0, 3, 3)), None, [])
    origin = np.stack([cols["ox"], cols["oy"], cols["oz"]], axis=1).tolist() if n else []
    Rl = R.tolist()
    inten, length = np.asarray(cols["intensity"], dtype=float).tolist(), np.asarray(cols["length"], dtype=float).tolist()
    qr, qi = np.asarray(cols["q_re"], dtype=float).tolist(), np.asarray(cols["q_im"], dtype=float).tolist()
    index = np.asarray(cols["n"], dtype=float).tolist()
    has_q = np.broadcast_to(np.asarray(has_q, dtype=bool), (n,)).tolist()
    inf = float("inf")
    for r in range(n):
        q = complex(qr[r], qi[r]) if has_q[r] else None
        yield (math_str(str(origin[r])),
               _OPPOSITE_TEXT if opposite[r] else math_str(str(Rl[r])),
               inten[r] if inten[r] else "None",                      # get_attr_str: a falsy attribute prints as None
               "None" if (length[r] == inf or not length[r]) else length[r],
               math_str(str(q if q else "None")),
               math_str(str(index[r] if index[r] else "None")))


def write_rays_csv(filename, cols, has_q):
    print(f"Exporting rays to {filename} ...")
    with open(filename, "w", newline="") as fh:
        writer = csv.writer(fh)
        rows = rays_csv_rows(cols, has_q)
        first = next(rows, None)
        writer.writerow(HEADER if first is not None else [])
        if first is not None:
            writer.writerow(first)
            writer.writerows(rows)


def columns_of_rays(rays):
    """The writer's columns from a list of Ray objects (the object API keeps `table.rays`)."""
    n = len(rays)
    o = np.array([r.origin for r in rays], dtype=float).reshape(n, 3)
    d = np.array([r.direction for r in rays], dtype=float).reshape(n, 3)
    q = np.array([complex(r.qo) if r.qo is not None else 0j for r in rays], dtype=complex)
    cols = dict(ox=o[:, 0], oy=o[:, 1], oz=o[:, 2], dx=d[:, 0], dy=d[:, 1], dz=d[:, 2],
                intensity=np.array([r.intensity for r in rays], dtype=float),
                length=np.array([np.inf if r.length is None else r.length for r in rays], dtype=float),
                q_re=q.real, q_im=q.imag, n=np.array([r.n for r in rays], dtype=float))
    return cols, np.array([r.qo is not None for r in rays], dtype=bool)


def parse_rays_csv(filename):
    """Numbers back out of an exported CSV (tests, round trips): dict of arrays; `None` fields become nan
    (length: +inf).  Accepts files written by the reference and by this package alike."""
    def num(text):
        return float(text.replace("*10^", "e"))

    def vec(text):
        return [num(t) for t in text.replace("{", " ").replace("}", " ").split(",")]

    out = {k: [] for k in ("origin", "R", "intensity", "length", "q", "n")}
    with open(filename, newline="") as fh:
        reader = csv.reader(fh)
        header = next(reader, None)
        if header:
            assert tuple(header) == HEADER, header
        for row in reader:
            out["origin"].append(vec(row[0]))
            out["R"].append(np.array(vec(row[1])).reshape(3, 3))
            out["intensity"].append(np.nan if row[2] == "None" else num(row[2]))
            out["length"].append(np.inf if row[3] == "None" else num(row[3]))
            out["q"].append(complex(np.nan, np.nan) if row[4] == "None" else complex(row[4].replace("*10^", "e").replace("I", "j")))
            out["n"].append(np.nan if row[5] == "None" else num(row[5]))
    return {k: np.array(v) for k, v in out.items()}
