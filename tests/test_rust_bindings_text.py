"""The Rust `-sys` crate under bindings/rust cannot be compiled here (no Rust toolchain), so this keeps
its `extern "C"` block in step with include/eigentrust_hip.h textually: same function names, same
argument counts, same error codes, same struct field counts."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "eigentrust_hip.h")
RUST = os.path.join(ROOT, "bindings", "rust", "eigentrust-hip-sys", "src", "lib.rs")


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _count_args(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return 0
    return arglist.count(",") + 1


def header_functions():
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"\b(et_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        out[m.group(1)] = _count_args(m.group(2))
    return out


def rust_functions():
    text = re.sub(r"//[^\n]*", " ", open(RUST).read())
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (et_[a-z0-9_]+)\s*\(([^()]*)\)", block):
        out[m.group(1)] = _count_args(m.group(2))
    return out


def test_same_functions_and_argument_counts():
    h, r = header_functions(), rust_functions()
    assert len(h) > 60
    assert sorted(h) == sorted(r)
    assert {k: (h[k], r[k]) for k in h if h[k] != r[k]} == {}


def test_same_error_codes():
    text = _strip_c_comments(open(HEADER).read())
    h = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(ET_(?:OK|E_[A-Z_]+))\s*=\s*(-?\d+)", text)}
    r = {m.group(1): int(m.group(2))
         for m in re.finditer(r"pub const (ET_(?:OK|E_[A-Z_]+)): c_int = (-?\d+);", open(RUST).read())}
    assert len(h) == 23 and h == r


def test_same_struct_field_counts():
    text = _strip_c_comments(open(HEADER).read())
    rust = re.sub(r"//[^\n]*", " ", open(RUST).read())
    for name in ("et_build_opts", "et_graph_info", "et_slice_desc", "et_work_desc", "et_plan_arrays",
                 "et_iter_stats", "et_score", "et_attestation"):
        body = re.search(r"typedef struct\s*\{([^{}]*)\}\s*%s\s*;" % name, text).group(1)
        n_c = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                n_c += decl.count(",") + 1
        rbody = re.search(r"pub struct %s\s*\{([^{}]*)\}" % name, rust).group(1)
        n_r = len(re.findall(r"pub [a-z0-9_]+\s*:", rbody))
        assert n_c == n_r, (name, n_c, n_r)
