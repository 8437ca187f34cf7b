"""Generator of tests/golden/reference_expectations.json (run in the build container, where /root/reference exists).

Reads the reference's own unit tests for the hot path and turns their expectations into DATA: inputs and expected
outputs, nothing else.  The reference states its expectations as closed-form C++ expressions next to literal inputs
(`EXPECT_THAT(pop_model.pop_integral(-5.0, 1.0), testing::DoubleNear(<expr>, 1e-6))`); this script evaluates those
expressions (they only use +-*/, std::exp / log / expm1 / log1p and local `auto x = ...;` definitions) and records
the numbers.  Covered:

  * tests/pop_model_tests.cpp          every EXPECT on pop_at_time / pop_integral / intensity_integral / log_N of the
                                       Const, Exp (+ min_pop) and Skygrid (staircase, log-linear) models, and the
                                       constructor arguments the reference rejects (std::invalid_argument);
  * tests/interval_set_tests.cpp       inserts, contains, merge, intersect, subtract, is_subset_of, slow_elements;
  * tests/scalable_coalescent_tests.cpp  the `log_prior` case: node times, tip flags and the three expected priors;
  * tests/phylo_tree_tests.cpp         the fixture tree's topology and times, and the find_MRCA_of / descends_from tables over its nodes.

Not covered here (kept as C++ in oracle/orc_tests.cpp, which restates the fixtures): tests that build trees
(tree_editing, spr_study, spr_move, phylo_tree_calc, missation_map, site_deltas), printing and derivative tests (off the path).
Usage: python tests/golden/make_reference_expectations.py [/root/reference]"""
import json
import math
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_expectations.json")


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def test_blocks(path):
    """[(test name, first line number, body text)] of a gtest file."""
    src = open(path).read()
    out = []
    for m in re.finditer(r"^TEST(?:_F)?\((\w+),\s*(\w+)\)\s*\{", src, flags=re.M):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        out.append((m.group(2), src[: m.start()].count("\n") + 1, strip_comments(src[m.end(): i - 1])))
    return out


def statements(body):
    """Top-level statements of a block (split at ';' outside brackets)."""
    out, depth, cur = [], 0, ""
    for ch in body:
        depth += {"(": 1, "{": 1, "[": 1, ")": -1, "}": -1, "]": -1}.get(ch, 0)
        if ch == ";" and depth == 0:
            out.append(" ".join(cur.split())); cur = ""
        else:
            cur += ch
    return [s for s in out if s]


FUNCS = {"exp": math.exp, "log": math.log, "expm1": math.expm1, "log1p": math.log1p, "sqrt": math.sqrt, "pow": math.pow, "abs": abs}


def cxx_eval(expr, env):
    e = expr.replace("std::numbers::ln2", repr(math.log(2.0)))
    e = re.sub(r"std::(\w+)", r"\1", e)
    e = re.sub(r"(\d)\.(?=[^\d]|$)", r"\1.0", e)           # "5." -> "5.0"
    return float(eval(e, {"__builtins__": {}}, {**FUNCS, **env}))


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += {"(": 1, "{": 1, ")": -1, "}": -1}.get(ch, 0)
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_model(ctor, env):
    m = re.match(r"(Const_pop_model|Exp_pop_model|Skygrid_pop_model)\s*\{(.*)\}$", ctor.strip(), flags=re.S)
    if not m:
        return None
    kind, args = m.group(1), split_args(m.group(2))
    if kind == "Const_pop_model":
        return {"kind": "const", "pop": cxx_eval(args[0], env)}
    if kind == "Exp_pop_model":
        v = [cxx_eval(a, env) for a in args]
        return {"kind": "exp", "t0": v[0], "n0": v[1], "g": v[2], "min_pop": v[3]}
    xs = [cxx_eval(a, env) for a in split_args(args[0].strip()[1:-1])] if args[0].strip() != "{}" else []
    gs = [cxx_eval(a, env) for a in split_args(args[1].strip()[1:-1])] if args[1].strip() != "{}" else []
    return {"kind": "skygrid", "x": xs, "gamma": gs, "type": "staircase" if "k_staircase" in args[2] else "log_linear"}


def pop_model_cases():
    path = os.path.join(REF, "tests", "pop_model_tests.cpp")
    cases, skipped = [], 0
    for name, line, body in test_blocks(path):
        env, models = {}, {}
        for st in statements(body):
            m = re.match(r"auto (\w+) = (.*)$", st)
            if m:
                mod = parse_model(m.group(2), env)
                if mod is not None:
                    models[m.group(1)] = mod
                    continue
                try:
                    env[m.group(1)] = cxx_eval(m.group(2), env)
                except Exception:
                    pass
                continue
            m = re.match(r"EXPECT_THROW\(\((\w+\{.*\})\), std::invalid_argument\)$", st) or re.match(r"EXPECT_THAT\(\(\[&\] \{ (\w+\{.*\}); \}\), testing::Throws<std::invalid_argument>\(\)\)$", st)
            if m:
                try:
                    mod = parse_model(m.group(1), env)
                except Exception:
                    mod = None
                if mod is not None:
                    cases.append({"test": name, "model": mod, "op": "construct", "expect": "invalid_argument"})
                    continue
            m = re.match(r"EXPECT_THAT\((\w+)\.(pop_at_time|pop_integral|intensity_integral|log_N)\((.*?)\), testing::DoubleNear\((.*), ([0-9.e+-]+)\)\)$", st) or \
                re.match(r"EXPECT_EQ\((\w+)\.(pop_at_time|pop_integral|intensity_integral|log_N)\((.*?)\), (.*)()\)$", st)
            if m and m.group(1) in models:
                try:
                    args = [cxx_eval(a, env) for a in split_args(m.group(3))]
                    expected = cxx_eval(m.group(4), env)
                except Exception:
                    skipped += 1
                    continue
                cases.append({"test": name, "model": models[m.group(1)], "op": m.group(2), "args": args, "expected": expected,
                              "tol": float(m.group(5)) if m.group(5) else 0.0})
                continue
            if st.startswith("EXPECT"):
                skipped += 1
    return cases, skipped


def parse_intervals(s):
    return [[int(a), int(b)] for a, b in re.findall(r"Site_interval\{\s*(\d+),\s*(\d+)\s*\}", s)]


def interval_set_cases():
    path = os.path.join(REF, "tests", "interval_set_tests.cpp")
    cases, skipped = [], 0
    for name, line, body in test_blocks(path):
        sets = {}
        for st in statements(body):
            m = re.match(r"auto (\w+) = (?:Scratch_interval_set|Interval_set<>)\{(.*)\}$", st)
            if m:
                sets[m.group(1)] = {"initial": parse_intervals(m.group(2)), "inserts": []}
                continue
            m = re.match(r"(?:auto \[\w+, \w+\] = )?(\w+)\.insert\((.*)\)$", st)
            if m and m.group(1) in sets:
                a = re.findall(r"\d+", m.group(2))
                sets[m.group(1)]["inserts"].append([int(a[0]), int(a[1])] if len(a) == 2 else [int(a[0]), int(a[0]) + 1])
                continue
            m = re.match(r"(\w+) = (\w+)$", st)
            if m and m.group(1) in sets and m.group(2) in sets:
                sets[m.group(1)] = json.loads(json.dumps(sets[m.group(2)]))
                continue
            m = re.match(r"EXPECT_THAT\((\w+), testing::ElementsAre\((.*)\)\)$", st)
            if m and m.group(1) in sets:
                cases.append({"test": name, "op": "elements", "set": json.loads(json.dumps(sets[m.group(1)])), "expected": parse_intervals(m.group(2))})
                continue
            m = re.match(r"EXPECT_THAT\((\w+)\.contains\((\d+)\), testing::(IsTrue|IsFalse)\(\)\)$", st)
            if m and m.group(1) in sets:
                cases.append({"test": name, "op": "contains", "set": json.loads(json.dumps(sets[m.group(1)])), "site": int(m.group(2)), "expected": m.group(3) == "IsTrue"})
                continue
            m = re.match(r"EXPECT_THAT\((merge_interval_sets|intersect_interval_sets|subtract_interval_sets)\((\w+), (\w+)\), testing::ElementsAre\((.*)\)\)$", st)
            if m and m.group(2) in sets and m.group(3) in sets:
                cases.append({"test": name, "op": m.group(1).split("_")[0], "a": json.loads(json.dumps(sets[m.group(2)])), "b": json.loads(json.dumps(sets[m.group(3)])), "expected": parse_intervals(m.group(4))})
                continue
            m = re.match(r"EXPECT_THAT\((interval_set_is_subset_of|interval_sets_intersect)\((\w+), (\w+)\), testing::(IsTrue|IsFalse)\(\)\)$", st)
            if m and m.group(2) in sets and m.group(3) in sets:
                cases.append({"test": name, "op": "is_subset_of" if "subset" in m.group(1) else "intersects", "a": json.loads(json.dumps(sets[m.group(2)])), "b": json.loads(json.dumps(sets[m.group(3)])), "expected": m.group(4) == "IsTrue"})
                continue
            m = re.match(r"EXPECT_THAT\(estd::ranges::to_vec\((\w+)\.slow_elements\(\)\), testing::ElementsAre\((.*)\)\)$", st)
            if m and m.group(1) in sets:
                cases.append({"test": name, "op": "slow_elements", "set": json.loads(json.dumps(sets[m.group(1)])), "expected": [int(x) for x in re.findall(r"\d+", m.group(2))]})
                continue
            if st.startswith("EXPECT"):
                skipped += 1
    return cases, skipped


def scalable_coalescent_case():
    """tests/scalable_coalescent_tests.cpp `log_prior`: three stages of displacements and the expected prior after each."""
    path = os.path.join(REF, "tests", "scalable_coalescent_tests.cpp")
    name, line, body = [b for b in test_blocks(path) if b[0] == "log_prior"][0]
    env = {}
    stages, coal, tips = [], {}, {}
    # the tip displacements are written as a range-for: make it one statement of its own
    body = re.sub(r"for \(const auto& t : \{([^}]*)\}\) \{\s*prior\.displace_tip\(i, t\);\s*\+\+i;\s*\}", r"DISPLACE_TIPS(\1);", body)
    body = re.sub(r"for \([^)]*\) \{ prior\.mark_as_\w+\(i\); \}", "", body)   # nodes 0 .. num_tips - 2 are coalescences, the rest tips
    for st in statements(body):
        m = re.match(r"auto (\w+) = (.*)$", st)
        if m and not m.group(2).startswith(("std::make_shared", "Scalable_coalescent_prior")):
            try:
                env[m.group(1)] = cxx_eval(m.group(2), env)
            except Exception:
                pass
            continue
        m = re.match(r"(\w+) = (0\.0.*)$", st)
        if m:
            env[m.group(1)] = cxx_eval(m.group(2), env)
            continue
        m = re.match(r"prior\.displace_coalescence\((\d+), (-?\d+)\)$", st)
        if m:
            coal[int(m.group(1))] = float(m.group(2))
            continue
        m = re.match(r"DISPLACE_TIPS\((.*)\)$", st)
        if m:
            for k, v in enumerate(split_args(m.group(1))):
                tips[int(env["num_tips"]) - 1 + k] = float(v)
            continue
        m = re.match(r"EXPECT_NEAR\(prior\.calc_log_prior\(\), (.*), ([0-9.e+-]+)\)$", st)
        if m:
            stages.append({"coalescence_times": dict(coal), "tip_times": dict(tips), "expected_log_prior": cxx_eval(m.group(1), env), "tol": float(m.group(2))})
    return {"test": name, "pop": env["pop"], "num_tips": int(env["num_tips"]), "t_ref": 0.0, "t_step": 1.0, "stages": stages}


def phylo_tree_query_cases():
    """tests/phylo_tree_tests.cpp: the fixture tree of Phylo_tree_complex_test (topology and node times only) and the tables of
    find_MRCA_of(tree, P, Q) / descends_from(tree, X, A) over pairs of its nodes and k_no_node (-1), with the fixture's times
    and with every time set to 0."""
    path = os.path.join(REF, "tests", "phylo_tree_tests.cpp")
    src = strip_comments(open(path).read())
    fx = src[src.index("class Phylo_tree_complex_test"):]
    fx = fx[: fx.index("\n};") + 3]
    idx = {m.group(1): int(m.group(2)) for m in re.finditer(r"static constexpr Node_index (\w+) = (\d+);", fx)}
    idx["k_no_node"] = -1
    n = len(idx) - 1
    parent, kids, t = [-1] * n, [[-1, -1] for _ in range(n)], [0.0] * n
    for m in re.finditer(r"tree\.at\((\w+)\)\.parent = (\w+);", fx): parent[idx[m.group(1)]] = idx[m.group(2)]
    for m in re.finditer(r"tree\.at\((\w+)\)\.children = \{(\w+), (\w+)\};", fx): kids[idx[m.group(1)]] = [idx[m.group(2)], idx[m.group(3)]]
    for m in re.finditer(r"tree\.at\((\w+)\)\.t = (?:tree\.at\(\w+\)\.t_min = tree\.at\(\w+\)\.t_max = )?(-?[0-9.]+);", fx): t[idx[m.group(1)]] = float(m.group(2))
    root = idx[re.search(r"tree\.root = (\w+);", fx).group(1)]
    blocks = {b[0]: b for b in test_blocks(path)}
    def table(test, fn, pat_expect):
        out = []
        for m in re.finditer(r"EXPECT_THAT\(%s\(tree, (\w+), (\w+)\), %s\)" % (fn, pat_expect), blocks[test][2]):
            e = m.group(3)
            out.append([idx[m.group(1)], idx[m.group(2)], idx[e] if e in idx else int(e == "true")])
        return out
    mrca = table("find_MRCA_of_nodes", "find_MRCA_of", r"testing::Eq\((\w+)\)")
    mrca_eq = table("find_MRCA_of_nodes_all_times_equal", "find_MRCA_of", r"testing::Eq\((\w+)\)")
    desc = table("descends_from_nodes", "descends_from", r"(true|false)")
    assert len(mrca) == 36 and len(mrca_eq) == 36 and len(desc) == 36, (len(mrca), len(mrca_eq), len(desc))
    return {"names": {k: v for k, v in idx.items()}, "root": root, "parent": parent, "children": kids, "t": t,
            "find_MRCA_of": mrca, "find_MRCA_of_all_times_equal": mrca_eq, "descends_from": desc}


if __name__ == "__main__":
    pop, sk1 = pop_model_cases()
    iv, sk2 = interval_set_cases()
    sc = scalable_coalescent_case()
    out = {"source": "expectations of the reference's tests/pop_model_tests.cpp, interval_set_tests.cpp, scalable_coalescent_tests.cpp, evaluated by tests/golden/make_reference_expectations.py",
           "pop_model": pop, "interval_set": iv, "scalable_coalescent": sc, "phylo_tree_queries": phylo_tree_query_cases(),
           "not_converted": {"pop_model_tests.cpp": sk1, "interval_set_tests.cpp": sk2, "why": "accessors, printing, iterator-identity and derivative expectations (off the hot path)"}}
    json.dump(out, open(OUT, "w"), indent=0)
    print("pop_model cases %d (skipped %d) | interval_set cases %d (skipped %d) | scalable_coalescent stages %d" % (len(pop), sk1, len(iv), sk2, len(sc["stages"])))
    from collections import Counter
    print(Counter(c["test"] for c in pop)); print(Counter(c["test"] for c in iv))
