"""The Rcpp shims under r-glue/src cannot be compiled here (no R toolchain), so they are checked as
text: each .Call routine of the hot path (/root/reference/src/init.cpp:9-35, names and arities
restated below) has exactly one shim with that arity, and every sarlacc_* call in the shims names a
function that include/sarlacc_amd.h declares, with the declared number of arguments."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "r-glue", "src")

# routine -> number of SEXP arguments (init.cpp:9-35): all 18 registered routines
ROUTINES = {
    "adaptor_align": 8, "adaptor_align_score_only": 6, "barcode_align": 6, "general_align": 7,
    "mask_bad_bases": 4, "unmask_alignment": 2,
    "create_consensus_basic": 3, "create_consensus_basic_loop": 3,
    "create_consensus_quality": 4, "create_consensus_quality_loop": 4,
    "umi_group": 5, "fast_levdist_test": 3, "cluster_umis_test": 1, "compute_lev_masked": 1,
    "quick_msa": 7,
    "find_homopolymers": 1, "match_homopolymers": 2, "find_errors": 2,
}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _split_args(s):
    """top-level comma split of an argument list"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _calls(text, prefix):
    """(name, [args]) for every `prefix...(` occurrence, with balanced parentheses"""
    res = []
    for m in re.finditer(r"\b(" + prefix + r"\w*)\s*\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        res.append((m.group(1), _split_args(text[m.end():i - 1])))
    return res


def _header_decls():
    text = _strip_comments(open(os.path.join(ROOT, "include", "sarlacc_amd.h")).read())
    decls = {}
    for name, args in _calls(text, "sarlacc_"):
        decls[name] = 0 if args == ["void"] else len(args)
    return decls


def _glue_sources():
    return {f: _strip_comments(open(os.path.join(GLUE, f)).read()) for f in sorted(os.listdir(GLUE))
            if f.endswith((".cpp", ".h"))}


def test_every_routine_has_one_shim_with_the_registered_arity():
    found = {}
    for fname, text in _glue_sources().items():
        for m in re.finditer(r"^SEXP\s+(\w+)\s*\(([^)]*)\)\s*\{", text, flags=re.M):
            args = _split_args(m.group(2))
            assert all(a.startswith("SEXP") for a in args), (fname, m.group(1))
            assert m.group(1) not in found, "two shims for " + m.group(1)
            found[m.group(1)] = len(args)
    assert found == ROUTINES


def test_every_abi_call_matches_the_header():
    decls = _header_decls()
    used = set()
    for fname, text in _glue_sources().items():
        for name, args in _calls(text, "sarlacc_"):
            assert name in decls, "%s calls %s, which include/sarlacc_amd.h does not declare" % (fname, name)
            assert len(args) == decls[name], "%s: %s called with %d arguments, declared with %d" % (
                fname, name, len(args), decls[name])
            used.add(name)
    # one host-pointer entry point per routine family
    expect = {"sarlacc_adaptor_align", "sarlacc_adaptor_align_score_only", "sarlacc_barcode_align", "sarlacc_general_align",
              "sarlacc_mask_bad_bases", "sarlacc_unmask_alignment", "sarlacc_umi_group", "sarlacc_fast_levdist_test",
              "sarlacc_cluster_umis_test", "sarlacc_compute_lev_masked", "sarlacc_quick_msa",
              "sarlacc_create_consensus_basic_loop", "sarlacc_create_consensus_quality_loop", "sarlacc_last_error",
              "sarlacc_find_homopolymers", "sarlacc_match_homopolymers", "sarlacc_find_errors"}
    assert expect <= used


def test_shims_wrap_their_bodies_for_rcpp_and_check_status():
    for fname, text in _glue_sources().items():
        if not fname.endswith(".cpp"):
            continue
        assert text.count("BEGIN_RCPP") == text.count("END_RCPP") == len(re.findall(r"^SEXP\s+\w+\s*\(", text, flags=re.M)), fname
        # a compute call outside SL_CHECK must be the guarded first attempt of a sizing protocol
        for m in re.finditer(r"^(.*)\bsarlacc_(?!last_error)\w+\s*\(", text, flags=re.M):
            assert "SL_CHECK" in m.group(1) or m.group(1).strip().startswith("if ("), (fname, m.group(0))


def test_makevars_links_the_library():
    mk = open(os.path.join(GLUE, "Makevars")).read()
    assert "-lsarlacc_amd" in mk and "-I$(SARLACC_AMD_HOME)/include" in mk


# the four scalar checkers of the reference package's utils.h that the shims call: name -> C++ return type
CHECKERS = {"check_integer_scalar": "int", "check_numeric_scalar": "double", "check_logical_scalar": "bool", "check_string": "std::string"}


def _write_package_headers(where):
    """sarlacc.h and utils.h of the shimmed package stay in that package; for the syntax check their declarations are written
    here from the tables above (routine -> arity; checker -> return type), nothing is kept in the tree."""
    with open(os.path.join(where, "sarlacc.h"), "w") as fh:
        fh.write('#pragma once\n#include "Rcpp.h"\nextern "C" {\n')
        for name, arity in ROUTINES.items():
            fh.write("SEXP %s(%s);\n" % (name, ", ".join(["SEXP"] * arity)))
        fh.write("}\n")
    with open(os.path.join(where, "utils.h"), "w") as fh:
        fh.write('#pragma once\n#include "Rcpp.h"\n#include <string>\n')
        for name, ret in CHECKERS.items():
            fh.write("%s %s(Rcpp::RObject, const char*);\n" % (ret, name))


def test_shims_parse_and_type_check(tmp_path):
    """Every shim goes through the compiler's front end (g++ -fsyntax-only) against declaration-only stand-ins for the
    headers it includes (tests/rglue_stubs: the shapes of the Rcpp / Biostrings calls the shims make; the shimmed package's
    own sarlacc.h / utils.h are generated into tmp_path from ROUTINES / CHECKERS) and the REAL include/sarlacc_amd.h: a typo,
    a missing argument or a wrong pointer type inside a shim body fails here, which the text checks above cannot see.
    Nothing is built or linked."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    stubs = os.path.join(ROOT, "tests", "rglue_stubs")
    gen = str(tmp_path)
    _write_package_headers(gen)
    inc = ["-I", gen, "-I", stubs, "-I", os.path.join(ROOT, "include"), "-I", GLUE]
    shims = sorted(f for f in os.listdir(GLUE) if f.endswith(".cpp"))
    assert len(shims) >= 13
    for f in shims:
        res = subprocess.run([gxx, "-std=c++14", "-fsyntax-only", "-Wall", "-Werror=return-type"] + inc + [os.path.join(GLUE, f)],
                             capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, "%s:\n%s" % (f, res.stderr[-3000:])
    # and the check bites: a shim with a wrong argument type does not pass
    bad = os.path.join(gen, "_bad_probe.cpp")
    with open(bad, "w") as fh:
        fh.write('#include "sarlacc.h"\n#include "utils.h"\n#include "flatten.h"\n'
                 'SEXP mask_bad_bases(SEXP a, SEXP b, SEXP c, SEXP d) { BEGIN_RCPP Flat s = flatten(a, true); '
                 'return Rcpp::List::create(sarlacc_mask_bad_bases(s.off.data())); END_RCPP }\n')
    res = subprocess.run([gxx, "-std=c++14", "-fsyntax-only"] + inc + [bad], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0
