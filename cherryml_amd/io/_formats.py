"""Readers / writers of the count-matrix and rate-matrix text files.

File grammar (reference: cherryml/io/_count_matrices.py:8-81):
    "<B> matrices\\n<S> states\\n" then, B times:
    "<q>\\n" "<state names, whitespace separated (leading blank)>\\n"
    S rows "<state> v_1 ... v_S" (tab or space separated).
Rate / mask / probability files are whitespace tables with a header row of
state names and the state name as first column (cherryml/io/_rate_matrix.py).

The parsers tokenise with numpy instead of per-line Python loops: the
co-evolution count file is ~80 MB of text.
"""
import os
from typing import List, Tuple

import numpy as np
import pandas as pd


def read_count_matrices_arrays(path: str) -> Tuple[np.ndarray, np.ndarray, List[str]]:
    """-> (q[B], C[B,S,S], states) as float64 arrays."""
    with open(path, "rb") as f:     # bytes: the native parser takes them as they are (no decode / encode of 84 MB)
        head1 = f.readline().decode("utf-8").strip().split(" ")
        head2 = f.readline().decode("utf-8").strip().split(" ")
        body = f.read()
    if len(head1) != 2 or head1[1] != "matrices":
        raise Exception(f"In file {path}, expected line '[num_matrices] matrices', "
                        f"but found: '{' '.join(head1)}'")
    if len(head2) != 2 or head2[1] != "states":
        raise Exception(f"In file {path}, expected line '[num_states] states', "
                        f"but found: '{' '.join(head2)}'")
    B, S = int(head1[0]), int(head2[0])
    native = _parse_native(path, body, B, S)
    if native is not None:
        return native
    tok = body.decode("utf-8").split()
    per = 1 + S + S * (S + 1)
    if len(tok) != B * per:
        raise Exception(f"Error reading count matrices file: {path}\nExpected {B} blocks of "
                        f"{S} states ({B * per} tokens), found {len(tok)} tokens")
    tok = np.asarray(tok, dtype=object).reshape(B, per)
    states = [str(s) for s in tok[0, 1:1 + S]]
    q = tok[:, 0].astype(np.float64)
    grid = tok[:, 1 + S:].reshape(B, S, S + 1)
    for b in range(B):
        if list(tok[b, 1:1 + S]) != states or list(grid[b, :, 0]) != states:
            raise Exception(f"Error reading count matrices file: {path}: state labels of "
                            f"matrix {b} differ from the first matrix")
    C = grid[:, :, 1:].astype(np.float64)
    return q, C, states


def _parse_native(path: str, raw: bytes, B: int, S: int):
    """The body through libcherrybank's host-side parser (cb_parse_count_matrices: all host threads, exact
    decimal -> double conversion; no GPU needed).  None when the library is not there (then the Python
    tokeniser below does the same job, ~30x slower at 400 states)."""
    try:
        from .. import _lib
        lib = _lib.load()
    except Exception:
        return None
    q = np.empty(B, dtype=np.float64)
    C = np.empty((B, S, S), dtype=np.float64)
    off = np.zeros(S, dtype=np.int64)
    ln = np.zeros(S, dtype=np.int32)
    rc = lib.cb_parse_count_matrices(raw, len(raw), B, S, q.ctypes.data, C.ctypes.data, off.ctypes.data,
                                     ln.ctypes.data, 0)
    if rc != 0:
        msg = lib.cb_last_error()
        msg = msg.decode() if isinstance(msg, bytes) else str(msg)
        raise Exception(f"Error reading count matrices file: {path}\n{msg}")
    states = [raw[o:o + n].decode("utf-8") for o, n in zip(off, ln)]
    return q, C, states


def read_count_matrices(path: str) -> List[Tuple[float, pd.DataFrame]]:
    """Same return type as the reference: list of (q, DataFrame S x S)."""
    q, C, states = read_count_matrices_arrays(path)
    return [(float(q[b]), pd.DataFrame(C[b], index=states, columns=states))
            for b in range(len(q))]


def _format_rows_native(M: np.ndarray, labels: List[str]):
    """bytes of "".join(label + "\t" + "\t".join(map(repr, row)) + "\n") through libcherrybank
    (cb_format_matrix_rows); None when the library is not there."""
    try:
        from .. import _lib
        lib = _lib.load()
    except Exception:
        return None
    import ctypes
    M = np.ascontiguousarray(M, dtype=np.float64)
    rows, cols = M.shape
    enc = [str(x).encode("utf-8") for x in labels]
    blob = b"".join(enc)
    ln = np.array([len(e) for e in enc], dtype=np.int32)
    off = np.zeros(rows, dtype=np.int64)
    if rows > 1:
        off[1:] = np.cumsum(ln[:-1])
    cap = rows * (int(ln.max(initial=0)) + 2 + 26 * cols) + 16
    out = ctypes.create_string_buffer(cap)
    written = ctypes.c_size_t(0)
    rc = lib.cb_format_matrix_rows(M.ctypes.data, rows, cols, blob, off.ctypes.data, ln.ctypes.data,
                                   ctypes.addressof(out), cap, ctypes.addressof(written))
    if rc != 0:
        return None
    return out.raw[:written.value]


def write_count_matrices(count_matrices: List[Tuple[float, pd.DataFrame]], path: str) -> None:
    d = os.path.dirname(path)
    if d != "" and not os.path.exists(d):
        os.makedirs(d)
    B = len(count_matrices)
    S = len(count_matrices[0][1])
    with open(path, "wb") as out:
        out.write(f"{B} matrices\n{S} states\n".encode("utf-8"))
        for q, m in count_matrices:
            out.write(f"{q}\n".encode("utf-8"))
            cols = [str(c) for c in m.columns]
            out.write(("\t" + "\t".join(cols) + "\n").encode("utf-8"))
            vals = np.asarray(m.to_numpy(), dtype=np.float64)
            body = _format_rows_native(vals, list(m.index))       # the bytes of repr(), formatted natively
            if body is None:    # Python floats: repr = shortest round trip
                body = "".join(str(name) + "\t" + "\t".join(map(repr, row)) + "\n"
                               for name, row in zip(m.index, vals.tolist())).encode("utf-8")
            out.write(body)


def _read_table(path: str) -> pd.DataFrame:
    return pd.read_csv(path, sep=r"\s+", index_col=0, keep_default_na=False, na_values=["_"],
                       float_precision="round_trip")


def read_rate_matrix(path: str) -> pd.DataFrame:
    return _read_table(path).astype(float)


def read_mask_matrix(path: str) -> pd.DataFrame:
    return _read_table(path).astype(int)


def read_probability_distribution(path: str) -> pd.DataFrame:
    res = _read_table(path).astype(float)
    if res.shape[1] != 1:
        raise Exception(f"Probability distribution at {path} should be one-dimensional.")
    if abs(res.sum().sum() - 1.0) > 1e-6:
        raise Exception(f"Probability distribution at {path} should add to 1.0, "
                        "with a tolerance of 1e-6.")
    return res


def write_rate_matrix(rate_matrix: np.ndarray, states: List[str], path: str) -> None:
    d = os.path.dirname(path)
    if d != "" and not os.path.exists(d):
        os.makedirs(d)
    M = np.asarray(rate_matrix)
    if M.dtype != np.float64 or M.ndim != 2 or np.isnan(M).any():
        pd.DataFrame(rate_matrix, index=states, columns=states).to_csv(path, sep="\t", index=True)
        return
    # the same bytes as DataFrame.to_csv (repr of every float), five times faster at 400 states -- the
    # co-evolution stage writes a dozen of these files (result, best, last, the 2^k snapshots)
    with open(path, "wb") as out:
        out.write(("\t" + "\t".join(str(c) for c in states) + "\n").encode("utf-8"))
        body = _format_rows_native(M, list(states))
        if body is None:
            body = "".join(str(name) + "\t" + "\t".join(map(repr, row)) + "\n"
                           for name, row in zip(states, M.tolist())).encode("utf-8")
        out.write(body)


def write_probability_distribution(p: np.ndarray, states: List[str], path: str) -> None:
    d = os.path.dirname(path)
    if d != "" and not os.path.exists(d):
        os.makedirs(d)
    if len(states) != p.shape[0]:
        raise Exception(f"probability_distribution has shape {p.shape}, "
                        f"inconsistent with states: {states}")
    df = pd.DataFrame(np.asarray(p).reshape(-1), index=states, columns=["prob"])
    df.index.name = "state"
    df.to_csv(path, sep="\t", index=True)
