"""The Rust shim (shim/src/*.rs) cannot be compiled in this image (no rustc / cargo): what CAN be checked without a compiler is that
every `datafusion::…` / `datafusion_ffi::…` item it imports exists, as a `pub` item, in the reference at the pinned version —
a renamed type, a moved module or a typo would otherwise only show when a maintainer first builds the shim.

Resolution is lexical: the umbrella crate's re-exports (datafusion/core/src/lib.rs:798-890: `pub mod physical_plan { pub use
datafusion_physical_plan::*; }` ...) map the first path segment to the source directories of the crates behind it; the module
segments must exist there as files / directories / `pub mod` declarations; the item must be declared `pub` (struct / enum /
trait / fn / type / const / static / mod / macro) or re-exported with `pub use` somewhere in those crates.  Skipped where
/root/reference is absent (the GPU box)."""
import os
import re

import pytest

REF = "/root/reference/datafusion"
SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shim", "src")

# first segment after `datafusion::` -> crate source directories that may hold the item (the alias's crate first, then what it re-exports)
CRATES = {
    "common": ["common/src", "common-runtime/src"],
    "config": ["common/src"],
    "error": ["common/src"],
    "catalog": ["catalog/src", "session/src"],
    "logical_expr": ["expr/src", "expr-common/src"],
    "physical_expr": ["physical-expr/src", "physical-expr-common/src"],
    "physical_expr_common": ["physical-expr-common/src"],
    "physical_optimizer": ["physical-optimizer/src", "session/src"],
    "physical_plan": ["physical-plan/src", "execution/src", "physical-expr/src", "physical-expr-common/src", "expr-common/src", "common/src"],
    "execution": ["core/src/execution", "execution/src", "session/src"],
    "physical_planner": ["core/src"],
    "datasource": ["core/src/datasource", "datasource/src", "datasource-parquet/src", "datasource-arrow/src", "catalog-listing/src"],
    "scalar": ["common/src"],
    "prelude": ["core/src"],
}
EXTERNAL = {"arrow", "parquet", "object_store"}   # third-party crates re-exported by the umbrella crate: not in /root/reference


def _imports():
    """every (file, full path) imported from datafusion / datafusion_ffi, `{a, b::{c, d}}` groups expanded"""
    out = []
    for fn in sorted(os.listdir(SHIM)):
        if not fn.endswith(".rs"):
            continue
        text = re.sub(r"//[^\n]*", "", open(os.path.join(SHIM, fn)).read())
        for m in re.finditer(r"\buse\s+((?:datafusion|datafusion_ffi)::[^;]+);", text):
            for p in _expand(re.sub(r"\s+", "", m.group(1))):
                out.append((fn, p))
    return out


def _expand(path):
    i = path.find("{")
    if i < 0:
        return [path]
    depth, j = 0, i
    for j in range(i, len(path)):
        depth += path[j] == "{"
        depth -= path[j] == "}"
        if depth == 0:
            break
    head, body, tail = path[:i], path[i + 1:j], path[j + 1:]
    parts, depth, cur = [], 0, ""
    for ch in body:
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            depth += ch == "{"
            depth -= ch == "}"
            cur += ch
    if cur:
        parts.append(cur)
    res = []
    for p in parts:
        res.extend(_expand(head + p + tail))
    return res


_SRC_CACHE = {}


def _sources(d):
    if d not in _SRC_CACHE:
        texts = []
        for root, _, files in os.walk(os.path.join(REF, d)):
            for f in files:
                if f.endswith(".rs"):
                    texts.append((os.path.join(root, f), open(os.path.join(root, f), errors="replace").read()))
        _SRC_CACHE[d] = texts
    return _SRC_CACHE[d]


def _declared_pub(item, dirs):
    decl = re.compile(r"\bpub(?:\([^)]*\))?\s+(?:unsafe\s+)?(?:async\s+)?(?:struct|enum|trait|fn|type|const|static|mod|union)\s+" + re.escape(item) + r"\b")
    reexp = re.compile(r"\bpub\s+use\s+[^;]*\b" + re.escape(item) + r"\b[^;]*;")
    macro = re.compile(r"macro_rules!\s+" + re.escape(item) + r"\b")
    for d in dirs:
        for _, text in _sources(d):
            if decl.search(text) or reexp.search(text) or macro.search(text):
                return True
    return False


def _module_exists(mod, dirs):
    for d in dirs:
        base = os.path.join(REF, d)
        for root, subdirs, files in os.walk(base):
            if mod + ".rs" in files or mod in subdirs:
                return True
        if any(re.search(r"\bpub\s+mod\s+" + re.escape(mod) + r"\b", t) for _, t in _sources(d)):
            return True
    return False


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not present on this machine")
def test_every_datafusion_item_the_shim_imports_is_public_in_the_reference():
    imports = _imports()
    assert len(imports) > 60, "the shim's imports were not found"
    missing = []
    for fn, path in imports:
        seg = path.split("::")
        if seg[0] == "datafusion_ffi":
            dirs, mods, item = ["ffi/src"], seg[1:-1], seg[-1]
        else:
            if seg[1] in EXTERNAL:
                continue
            if seg[1] not in CRATES:
                missing.append((fn, path, "unknown top-level module"))
                continue
            dirs, mods, item = CRATES[seg[1]], seg[2:-1], seg[-1]
        if item in ("*", "self"):
            continue
        item = item.split(" as ")[0]
        for mod in mods:
            if not _module_exists(mod, dirs):
                missing.append((fn, path, f"module {mod} not found under {dirs}"))
                break
        else:
            if not _declared_pub(item, dirs):
                missing.append((fn, path, f"no pub item {item} under {dirs}"))
    assert not missing, "\n".join(f"{f}: {p}: {why}" for f, p, why in missing)


def test_expand_handles_nested_groups():
    assert _expand("a::{b,c::{d,e},f}") == ["a::b", "a::c::d", "a::c::e", "a::f"]
