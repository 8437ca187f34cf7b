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
  * tests/phylo_tree_tests.cpp         the fixture tree's topology and times, and the find_MRCA_of / descends_from tables over its nodes;
  * tests/spr_move_tests.cpp           all six fixture trees with their evolution model, and every expectation of the nine
                                       analyze_graft_* tests (branch infos, partial lambdas, warm / hot sites, hot mutations and
                                       deltas, log_alpha_mut, delta_log_G -- the closed-form right-hand sides evaluated), of the
                                       tricky rooty graft's peel / closed-mutations / peel-and-reapply tests, the (X, SS, t) cases of
                                       the full_spr_move tests, and the parameters and expected frequencies of the
                                       sample_mutational_history test;
  * tests/phylo_tree_calc_tests.cpp    the fixture with its model, lambda_i and missing-site counts per node, log G below the root, the two
                                       root-prior cases with zero-probability states, calc_num_muts / _beta_ab / _l, calc_Ttwiddle_beta_a, calc_T;
  * tests/tree_editing_tests.cpp       all ten tests: the tree each test starts its editing session from, the session's steps, and the
                                       expected times, mutation lists, missations, parents and root afterwards.

Not covered here (kept as C++ in oracle/orc_tests.cpp, which restates the fixtures): the other tests that build trees
(spr_study, the rest of phylo_tree_calc, missation_map, site_deltas), printing and derivative tests (off the path).
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


# ---- tests/spr_move_tests.cpp ---------------------------------------------------------------------------------------------
STATE = {"rA": 0, "rC": 1, "rG": 2, "rT": 3}
NEG_DBL_MAX = -1.7976931348623157e308
FLT_MAX = 3.4028234663852886e38


def _block_after(src, start):
    """Text of the {...} block whose opening brace is the first '{' at or after `start`; returns (text, end index)."""
    i = src.index("{", start)
    depth, j = 1, i + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        j += 1
    return src[i + 1: j - 1], j


_NUM_ENV = {}      # "tree.at(r).t" -> value, while a test body of tree_editing_tests.cpp is being read


def _num(tok):
    tok = tok.strip()
    if tok in _NUM_ENV: return _NUM_ENV[tok]
    if tok in ("-std::numeric_limits<double>::max()",): return NEG_DBL_MAX
    if tok in ("-std::numeric_limits<float>::max()",): return -FLT_MAX
    if tok in ("+std::numeric_limits<float>::max()", "std::numeric_limits<float>::max()"): return FLT_MAX
    return float(tok)


def _mutations(text):
    """[[from, site, to, t]] of every Mutation{rX, site, rY, t} in `text`, in order."""
    return [[STATE[a], int(l), STATE[b], _num(t)] for a, l, b, t in re.findall(r"Mutation\{(r[ACGT]),\s*(\d+),\s*(r[ACGT]),\s*([^}]*)\}", text)]


def _missations(text):
    return [[int(l), STATE[a]] for l, a in re.findall(r"Missation\{(\d+),\s*(r[ACGT])\}", text)]


def _states(text):
    return [STATE[x] for x in re.findall(r"r[ACGT]", text)]


def _fixture_model(txt):
    """(reference sequence, partition_for_site, nu_l, mu[2], pi[2][4], q[2][4][4]) as a fixture class of the reference's tests sets them up."""
    ref = _states(re.search(r"Real_sequence ref_sequence\{([^}]*)\}", txt).group(1))
    pfs = [int(x) for x in re.search(r"make_global_evo_model\(\{([^}]*)\}\)", txt).group(1).split(",")]
    q = [[[0.0] * 4 for _ in range(4)] for _ in range(2)]
    for p_, a, b, e in re.findall(r"q_(\d)_ab\[(r[ACGT])\]\[(r[ACGT])\] = ([^;]*);", txt):
        q[int(p_)][STATE[a]][STATE[b]] = cxx_eval(e, {})
    nu = [float(x) for x in re.search(r"evo\.nu_l = \{([^}]*)\};", txt).group(1).split(",")]
    mu, pi = [0.0, 0.0], [None, None]
    for p_, body in re.findall(r"evo\.partition_evo_model\[(\d)\] = \{(.*?)\};", txt, flags=re.S):
        mu[int(p_)] = float(re.search(r"\.mu = ([0-9.]+)", body).group(1))
        pi[int(p_)] = [float(x) for x in re.search(r"\.pi_a = \{([^}]*)\}", body).group(1).split(",")]
    for p_ in range(2):
        for a in range(4):
            assert abs(sum(q[p_][a])) < 1e-12, "rows of a rate matrix sum to zero"
    return ref, pfs, nu, mu, pi, q


def _fixture_nodes(ctor_body, idx, n):
    nodes = [{"parent": -1, "children": [], "t": 0.0, "t_min": 0.0, "t_max": 0.0, "mutations": [], "missations": [], "name": ""} for _ in range(n)]
    for st in statements(ctor_body):
        mm = re.search(r"tree\.at\((\w+)\)\.(\w+) = (.*)$", st)
        if not mm: continue
        nd, field, val = nodes[idx[mm.group(1)]], mm.group(2), mm.group(3)
        if field == "parent": nd["parent"] = idx[val.strip()]
        elif field == "children": nd["children"] = [idx[x.strip()] for x in val.strip()[1:-1].split(",") if x.strip()]
        elif field == "name": nd["name"] = val.strip().strip('"')
        elif field == "mutations": nd["mutations"] = _mutations(val)
        elif field == "missations": nd["missations"] = _missations(val)
        elif field in ("t", "t_min", "t_max"):
            v = _num(val.split("=")[-1])
            for f in re.findall(r"\.(t_min|t_max)\b", val) + [field]: nd[f] = v
    return nodes


def _model_env(evo, extra=None):
    env = {"rA": 0, "rC": 1, "rG": 2, "rT": 3,
           "mu_l": lambda l: evo["mu"][evo["partition_for_site"][l]], "nu_l": lambda l: evo["nu_l"][l],
           "pi_l_a": lambda l, a: evo["pi"][evo["partition_for_site"][l]][a],
           "q_l_ab": lambda l, a, b: evo["q"][evo["partition_for_site"][l]][a][b],
           "q_l_a": lambda l, a: -evo["q"][evo["partition_for_site"][l]][a][a]}
    env.update(extra or {})
    return env


def phylo_tree_calc_cases():
    """tests/phylo_tree_calc_tests.cpp: the fixture (tree + model) and what the tests on the hot path expect of it -- lambda_i per node
    (calc_lambda_i), missing-site counts, log G below the root, the root prior with zero-probability states, and the global moves'
    sufficient statistics (calc_num_muts, _beta_ab, _l, calc_Ttwiddle_beta_a, calc_T)."""
    path = os.path.join(REF, "tests", "phylo_tree_calc_tests.cpp")
    src = strip_comments(open(path).read())
    name = "Phylo_tree_calc_complex_test"
    txt, _ = _block_after(src, src.index("class " + name))
    ref, pfs, nu, mu, pi, q = _fixture_model(txt)
    idx = {n: int(i) for n, i in re.findall(r"static constexpr Node_index (\w+) = (\d+);", txt)}; idx["k_no_node"] = -1
    n = int(re.search(r"Phylo_tree tree\{(\d+)\};", txt).group(1))
    ctor, _ = _block_after(txt, txt.index(name + "()"))
    evo = {"partition_for_site": pfs, "nu_l": nu, "mu": mu, "pi": pi, "q": q}
    fx = {"names": {k: v for k, v in idx.items() if k != "k_no_node"}, "root": idx[re.search(r"tree\.root = (\w+);", ctor).group(1)], "ref_sequence": ref,
          "nodes": _fixture_nodes(ctor, idx, n), "evo": evo}
    blocks = {b[0]: b for b in test_blocks(path)}
    out = {"fixture": fx}
    # calc_lambda_i: one DoubleNear per node
    lam = [None] * n
    for st in statements(blocks["calc_lambda_i"][2]):
        mm = re.match(r"EXPECT_THAT\(lambda_i\[(\w+)\], testing::DoubleNear\((.*), ([0-9.e+-]+)\)\)$", st)
        if mm and mm.group(1) in idx: lam[idx[mm.group(1)]] = cxx_eval(mm.group(2), _model_env(evo))
    assert all(v is not None for v in lam)
    out["lambda_i"] = {"line": blocks["calc_lambda_i"][1], "expected": lam, "tol": 1e-6}
    miss = [None] * n
    for st in statements(blocks["calc_num_sites_missing_at_every_node"][2]):
        mm = re.match(r"EXPECT_THAT\(result\[(\w+)\], testing::Eq\((\d+)\)\)$", st)
        if mm: miss[idx[mm.group(1)]] = int(mm.group(2))
    assert all(v is not None for v in miss)
    out["num_sites_missing"] = {"line": blocks["calc_num_sites_missing_at_every_node"][1], "expected": miss}
    # calc_log_G_below_root: `expected += <sum>` statements
    g = 0.0
    for st in statements(blocks["calc_log_G_below_root"][2]):
        mm = re.match(r"expected \+= (.*)$", st)
        if mm: g += cxx_eval(mm.group(1), _model_env(evo))
    out["log_G_below_root"] = {"line": blocks["calc_log_G_below_root"][1], "expected": g, "tol": 1e-6}
    # the two root-prior tests: the model's pi replaced, then a closed form or -inf
    rp = []
    for t in ("calc_log_root_prior_zero_p_is_ok", "calc_log_root_prior_zero_p_is_impossible"):
        pi2 = [list(pi[0]), list(pi[1])]
        exp = None
        for st in statements(blocks[t][2]):
            mm = re.match(r"evo\.partition_evo_model\[(\d)\]\.pi_a = Seq_vector\{([^}]*)\}$", st)
            if mm: pi2[int(mm.group(1))] = [float(x) for x in mm.group(2).split(",")]; continue
            mm = re.match(r"auto expected = (.*)$", st)
            if mm: exp = cxx_eval(mm.group(1), _model_env({**evo, "pi": pi2}))
            if re.match(r"EXPECT_THAT\(result, testing::Eq\(-std::numeric_limits<double>::infinity\(\)\)\)$", st): exp = "-inf"
        assert exp is not None
        rp.append({"test": t, "line": blocks[t][1], "pi": pi2, "expected": exp, "tol": 1e-6})
    out["log_root_prior"] = rp
    # sufficient statistics of the global moves
    out["num_muts"] = int(re.search(r"calc_num_muts\(tree\), testing::Eq\((\d+)\)", blocks["calc_num_muts"][2]).group(1))
    M = [[[0] * 4 for _ in range(4)] for _ in range(2)]
    for p_, a, b in re.findall(r"\+\+expected\[(\d)\]\[(r[ACGT])\]\[(r[ACGT])\]", blocks["calc_num_muts_beta_ab"][2]): M[int(p_)][STATE[a]][STATE[b]] += 1
    out["num_muts_beta_ab"] = M
    out["num_muts_l"] = [int(x) for x in re.search(r"Node_vector<int>\{(.*?)\}", blocks["calc_num_muts_l"][2], flags=re.S).group(1).split(",")]
    T = [[0.0] * 4 for _ in range(2)]
    for p_, a, e in re.findall(r"expected\[(\d)\]\[(r[ACGT])\] \+= ([^;]*);", blocks["calc_Ttwiddle_beta_a"][2]): T[int(p_)][STATE[a]] += cxx_eval(e, _model_env(evo))
    out["Ttwiddle_beta_a"] = {"expected": T, "tol": 1e-6}
    tt = 0.0
    for e in re.findall(r"expected \+= ([^;]*);", blocks["calc_T"][2]): tt += cxx_eval(e, {})
    out["T"] = tt
    assert out["num_muts"] == sum(out["num_muts_l"]) == sum(sum(sum(r) for r in m) for m in M)
    return out


def _apply_node_statement(st, nodes, idx):
    mm = re.search(r"^tree\.at\((\w+)\)\.(\w+) = (.*)$", st)
    if not mm: return False
    nd, field, val = nodes[idx[mm.group(1)]], mm.group(2), mm.group(3)
    if field == "parent": nd["parent"] = idx[val.strip()]
    elif field == "children": nd["children"] = [idx[x.strip()] for x in val.strip()[1:-1].split(",") if x.strip()]
    elif field == "name": nd["name"] = val.strip().strip('"')
    elif field == "mutations": nd["mutations"] = _mutations(val)
    elif field == "missations": nd["missations"] = _missations(val)
    elif field in ("t", "t_min", "t_max"):
        v = _num(val.split("=")[-1])
        for f in re.findall(r"\.(t_min|t_max)\b", val) + [field]: nd[f] = v
    else: return False
    return True


def tree_editing_cases():
    """tests/tree_editing_tests.cpp: all ten tests -- the fixture as the test leaves it before the editing session opens, the session's steps
    (slide_P_along_branch, hop_up, flip, hop_down), and every expectation on times, mutation lists, missations, parents and the root,
    with `old_tree` references resolved to values."""
    path = os.path.join(REF, "tests", "tree_editing_tests.cpp")
    src = strip_comments(open(path).read())
    base_txt, _ = _block_after(src, src.index("class Tree_editing_test_base"))
    ref, pfs, nu, mu, pi, q = _fixture_model(base_txt)
    evo = {"partition_for_site": pfs, "nu_l": nu, "mu": mu, "pi": pi, "q": q}
    fixtures = {}
    for m in re.finditer(r"class (Tree_editing_\w+_test) : public Tree_editing_test_base \{", src):
        name = m.group(1)
        txt, _ = _block_after(src, m.start())
        idx = {n: int(i) for n, i in re.findall(r"static constexpr Node_index (\w+) = (\d+);", txt)}; idx["k_no_node"] = -1
        n = int(re.search(r"Phylo_tree tree\{(\d+)\};", txt).group(1))
        r2 = re.search(r"Real_sequence ref_sequence\{([^}]*)\}", txt)
        ctor, _ = _block_after(txt, txt.index(name + "()"))
        fixtures[name] = {"idx": idx, "root": idx[re.search(r"tree\.root = (\w+);", ctor).group(1)], "ref_sequence": _states(r2.group(1)) if r2 else list(ref),
                          "nodes": _fixture_nodes(ctor, idx, n)}
    tests = []
    for m in re.finditer(r"^TEST_F\((\w+),\s*(\w+)\)\s*\{", src, flags=re.M):
        fixture, test = m.group(1), m.group(2)
        body, _ = _block_after(src, m.end() - 1)
        fx = fixtures[fixture]; idx = fx["idx"]
        nodes = json.loads(json.dumps(fx["nodes"]))
        old = json.loads(json.dumps(nodes))
        root = fx["root"]
        names = {v: k for k, v in idx.items()}
        def refresh_env():
            _NUM_ENV.clear()
            for k, v in idx.items():
                if v >= 0: _NUM_ENV["tree.at(%s).t" % k] = nodes[v]["t"]
        X, ops, expect = None, [], {"t": {}, "mutations": {}, "missations": {}, "parent": {}}
        session_open = False
        for st in statements(body):
            refresh_env()
            if not session_open and _apply_node_statement(st, nodes, idx): continue
            if st == "old_tree = tree": old = json.loads(json.dumps(nodes)); continue
            mm = re.match(r"auto edit = Tree_editing_session\{tree, (\w+),", st)
            if mm: X = idx[mm.group(1)]; session_open = True; start = json.loads(json.dumps(nodes)); continue
            mm = re.match(r"edit\.slide_P_along_branch\((.*)\)$", st)
            if mm:
                t = _num(mm.group(1)); ops.append(["slide", t])
                continue
            if st == "edit.hop_up()": ops.append(["hop_up"]); continue
            if st == "edit.flip()": ops.append(["flip"]); continue
            mm = re.match(r"edit\.hop_down\((\w+)\)$", st)
            if mm: ops.append(["hop_down", idx[mm.group(1)]]); continue
            if st == "edit.end()": continue
            mm = re.match(r"EXPECT_THAT\(tree\.at\((\w+)\)\.t, testing::(Eq|DoubleNear)\((.*)\)\)$", st)
            if mm:
                arg = split_args(mm.group(3))[0]
                mo = re.match(r"old_tree\.at\((\w+)\)\.t$", arg)
                expect["t"][str(idx[mm.group(1)])] = old[idx[mo.group(1)]]["t"] if mo else float(arg)
                continue
            mm = re.match(r"EXPECT_THAT\(tree\.at\((\w+)\)\.mutations, testing::(.*)\)$", st)
            if mm:
                node, matcher = str(idx[mm.group(1)]), mm.group(2)
                mo = re.match(r"Eq\(old_tree\.at\((\w+)\)\.mutations\)$", matcher)
                if mo: expect["mutations"][node] = {"ordered": old[idx[mo.group(1)]]["mutations"]}
                elif matcher.startswith("UnorderedElementsAre("): expect["mutations"][node] = {"unordered": _mutations(matcher)}
                elif matcher.startswith("ElementsAre("): expect["mutations"][node] = {"ordered": _mutations(matcher)}
                elif matcher.startswith("IsEmpty"): expect["mutations"][node] = {"ordered": []}
                else: raise ValueError(st)
                continue
            mm = re.match(r"EXPECT_THAT\((?:estd::ranges::to_vec\()?tree\.at\((\w+)\)\.missations(?:\.slow_elements\(ref_sequence\)\))?, testing::(.*)\)$", st)
            if mm:
                node, matcher = str(idx[mm.group(1)]), mm.group(2)
                mo = re.match(r"Eq\(old_tree\.at\((\w+)\)\.missations\)$", matcher)
                if mo: expect["missations"][node] = old[idx[mo.group(1)]]["missations"]
                elif matcher.startswith("ElementsAre("): expect["missations"][node] = _missations(matcher)
                elif matcher.startswith("IsEmpty"): expect["missations"][node] = []
                else: raise ValueError(st)
                continue
            mm = re.match(r"EXPECT_THAT\(tree\.at\((\w+)\)\.parent, testing::Eq\((\w+)\)\)$", st)
            if mm: expect["parent"][str(idx[mm.group(1)])] = idx[mm.group(2)]; continue
            mm = re.match(r"EXPECT_THAT\(tree\.root, testing::Eq\((\w+)\)\)$", st)
            if mm: expect["root"] = idx[mm.group(1)]; continue
            if st.startswith("EXPECT_THAT(num_sites_missing_at_every_node"): expect["derived_quantities_kept"] = True; continue
            if st.startswith("EXPECT") : raise ValueError("unconverted expectation in %s.%s: %s" % (fixture, test, st))
        _NUM_ENV.clear()
        assert X is not None and ops
        tests.append({"fixture": fixture, "test": test, "line": src[: m.start()].count("\n") + 1, "X": X, "ops": ops,
                      "tree": {"names": {k: v for k, v in idx.items() if v >= 0}, "root": root, "ref_sequence": fx["ref_sequence"], "nodes": start, "evo": evo},
                      "expect": expect})
    assert len(tests) == 10, len(tests)
    return tests


def spr_move_cases():
    path = os.path.join(REF, "tests", "spr_move_tests.cpp")
    raw = open(path).read()
    src = strip_comments(raw)
    # -- the base fixture: reference sequence, the two-partition model, mu_JC
    base_txt, _ = _block_after(src, src.index("class Spr_move_test_base"))
    base_ref, base_pfs, base_nu, mu, pi, q = _fixture_model(base_txt)
    mu_JC = float(re.search(r"double mu_JC = ([0-9.]+);", base_txt).group(1))
    # -- the fixture classes
    fixtures = {}
    for m in re.finditer(r"class (Spr_move_\w+_test) : public Spr_move_test_base \{", src):
        name = m.group(1)
        txt, _ = _block_after(src, m.start())
        idx = {n: int(i) for n, i in re.findall(r"static constexpr Node_index (\w+) = (\d+);", txt)}
        idx["k_no_node"] = -1
        n = int(re.search(r"Phylo_tree tree\{(\d+)\};", txt).group(1))
        ref, pfs, nu = list(base_ref), list(base_pfs), list(base_nu)
        mm = re.search(r"tree\.ref_sequence = ref_sequence = \{([^}]*)\};", txt)
        if mm: ref = _states(mm.group(1))
        mm = re.search(r"evo\.partition_for_site = \{([^}]*)\};", txt)
        if mm: pfs = [int(x) for x in mm.group(1).split(",")]
        mm = re.search(r"evo\.nu_l = \{([^}]*)\};", txt)
        if mm: nu = [float(x) for x in mm.group(1).split(",")]
        nodes = _fixture_nodes(_block_after(txt, txt.index(name + "()"))[0], idx, n)
        root = idx[re.search(r"tree\.root = (\w+);", txt).group(1)]
        fixtures[name] = {"names": {k: v for k, v in idx.items() if k != "k_no_node"}, "root": root, "ref_sequence": ref, "nodes": nodes,
                          "evo": {"partition_for_site": pfs, "nu_l": nu, "mu": mu, "pi": pi, "q": q}}
    # -- the tests
    BI_INDEX = {"Spr_graft::k_branch_info_P_X": 0, "Spr_graft::k_branch_info_P_S": 1, "Spr_graft::k_branch_info_S_P_X": 2}
    graft_tests, peel_tests, move_cases, history = [], [], [], None
    for m in re.finditer(r"^TEST_F\((\w+),\s*(\w+)\)\s*\{", src, flags=re.M):
        fixture, test = m.group(1), m.group(2)
        body, _ = _block_after(src, m.end() - 1)
        line = src[: m.start()].count("\n") + 1
        fx = fixtures[fixture]; idx = dict(fx["names"]); idx["k_no_node"] = -1
        evo = fx["evo"]
        env = _model_env(evo, {"mu_JC": mu_JC, "P_JC": lambda a, b, t: (1.0 + 3. / 4 * math.expm1(-4. / 3. * mu_JC * t)) if a == b else (-1. / 4. * math.expm1(-4. / 3. * mu_JC * t))})
        sts = statements(body)
        if test.startswith("analyze_graft") or test in ("peel_graft_X", "closed_mutations_graft_X", "peel_reapply_graft_X"):
            ccr = True
            for st in sts:
                mm = re.match(r"auto can_change_root = (true|false)$", st)
                if mm: ccr = mm.group(1) == "true"
            ctor = [st for st in sts if st.startswith("auto spr = Spr_move{")][0]
            third = split_args(ctor[ctor.index("{") + 1: ctor.rindex("}")])[2]
            can_change_root = ccr if third == "can_change_root" else (third == "true")
            X = idx[re.search(r"spr\.analyze_graft\((\w+)\)", body).group(1)]
        if test.startswith("analyze_graft"):
            alias, bis, out = {}, {}, {"fixture": fixture, "test": test, "line": line, "can_change_root": can_change_root, "X": X}
            skipped = 0
            for st in sts:
                mm = re.match(r"const auto& (\w+) = analysis\.branch_infos\[(.*)\]$", st)
                if mm:
                    k = BI_INDEX.get(mm.group(2)); k = int(mm.group(2)) if k is None else k
                    alias[mm.group(1)] = k; bis[k] = {"index": k}
                    continue
                mm = re.match(r"EXPECT_THAT\(analysis\.branch_infos, testing::SizeIs\((\d+)\)\)$", st)
                if mm: out["num_branch_infos"] = int(mm.group(1)); continue
                mm = re.match(r"EXPECT_THAT\(analysis\.(log_alpha_mut|delta_log_G), testing::DoubleNear\((.*), ([0-9.e+-]+)\)\)$", st)
                if mm: out[mm.group(1)] = cxx_eval(mm.group(2), env); out["tol"] = float(mm.group(3)); continue
                mm = re.match(r"EXPECT_THAT\((?:estd::ranges::to_vec\()?(\w+)\.(\w+?)(?:\.slow_elements\(\)\))?, testing::(.*)\)$", st)
                if mm and mm.group(1) in alias:
                    b, field, matcher = bis[alias[mm.group(1)]], mm.group(2), mm.group(3)
                    if field in ("A", "B"): b[field] = idx[re.match(r"Eq\((\w+)\)", matcher).group(1)]
                    elif field == "is_open": b[field] = matcher.startswith("IsTrue")
                    elif field == "T_to_X": b[field] = float(re.match(r"Eq\(([0-9.]+)\)", matcher).group(1))
                    elif field in ("partial_lambda_at_A", "partial_lambda_at_X"):
                        mm2 = re.match(r"DoubleNear\((.*), ([0-9.e+-]+)\)$", matcher)
                        b[field] = cxx_eval(mm2.group(1), env) if mm2 else float(re.match(r"Eq\(([0-9.]+)\)", matcher).group(1))
                    elif field in ("warm_sites", "hot_sites"):
                        c = b.setdefault(field, {})
                        if matcher.startswith("ElementsAre("): c["intervals"] = parse_intervals(matcher)
                        elif matcher.startswith("IsEmpty"): c["intervals"] = []
                        elif matcher.startswith("IsSupersetOf("): c.setdefault("contains", []).extend(int(x) for x in re.findall(r"\d+", matcher))
                        elif matcher.startswith("Contains("): c.setdefault("contains", []).extend(int(x) for x in re.findall(r"\d+", matcher))
                        elif matcher.startswith("Not(testing::Contains("): c.setdefault("not_contains", []).extend(int(x) for x in re.findall(r"\d+", matcher))
                        else: raise ValueError(st)
                    elif field == "hot_muts_to_X": b[field] = _mutations(matcher)
                    elif field == "hot_deltas_to_X": b[field] = [[int(l), STATE[a], STATE[c_]] for l, a, c_ in re.findall(r"site_deltas_entry\((\d+),\s*(r[ACGT]),\s*(r[ACGT])\)", matcher)]
                    else: raise ValueError(st)
                    continue
                if st.startswith("EXPECT"): raise ValueError("unconverted expectation in %s.%s: %s" % (fixture, test, st))
            out["branch_infos"] = [bis[k] for k in sorted(bis)]
            assert len(out["branch_infos"]) == out["num_branch_infos"] and "log_alpha_mut" in out and "delta_log_G" in out, (fixture, test)
            graft_tests.append(out)
        elif test in ("peel_graft_X", "closed_mutations_graft_X", "peel_reapply_graft_X"):
            out = {"fixture": fixture, "test": test, "line": line, "can_change_root": can_change_root, "X": X,
                   "mode": 2 if "spr.apply_graft(analysis)" in body else 1, "nodes": {}}
            for st in sts:
                mm = re.match(r"EXPECT_THAT\(tree\.at\((\w+)\)\.mutations, testing::(.*)\)$", st)
                if mm: out["nodes"].setdefault(str(idx[mm.group(1)]), {})["mutations_unordered"] = _mutations(mm.group(2)); continue
                mm = re.match(r"EXPECT_THAT\(estd::ranges::to_vec\(tree\.at\((\w+)\)\.missations\.slow_elements\(ref_sequence\)\), testing::(.*)\)$", st)
                if mm: out["nodes"].setdefault(str(idx[mm.group(1)]), {})["missations"] = _missations(mm.group(2)); continue
                mm = re.match(r"EXPECT_THAT\(tree\.ref_sequence, testing::Eq\(Real_sequence\{(.*)\}\)\)$", st)
                if mm: out["ref_sequence"] = _states(mm.group(1)); continue
                if re.match(r"EXPECT_THAT\(tree\.ref_sequence, testing::Eq\(ref_sequence\)\)$", st): continue
                mm = re.match(r"EXPECT_THAT\(spr\.count_closed_mutations\(analysis\), testing::Eq\((\d+)\)\)$", st)
                if mm: out["count_closed_mutations"] = int(mm.group(1)); continue
                mm = re.match(r"EXPECT_THAT\(spr\.summarize_closed_mutations\(analysis\), testing::UnorderedElementsAre\((.*)\)\)$", st)
                if mm: out["closed_deltas"] = [[int(l), STATE[a], STATE[c_]] for l, a, c_ in re.findall(r"site_deltas_entry\((\d+),\s*(r[ACGT]),\s*(r[ACGT])\)", mm.group(1))]; continue
                if st.startswith("EXPECT"): raise ValueError("unconverted expectation in %s.%s: %s" % (fixture, test, st))
            peel_tests.append(out)
        elif test == "full_spr_move":
            cases = [[idx[a], idx[b], float(t)] for a, b, t in re.findall(r"Case\{(\w+),\s*(\w+),\s*(-?[0-9.]+)\}", body)]
            seeds = int(re.search(r"seed != (\d+)", body).group(1))
            move_cases.append({"fixture": fixture, "line": line, "cases": cases, "seeds": seeds})
        elif test == "sample_mutational_history":
            history = {"fixture": fixture, "line": line, "num_histories": int(re.search(r"is_debug_enabled \? [0-9']+ : ([0-9']+)", body).group(1).replace("'", "")),
                       "target_start_seq": _states(re.search(r"target_start_seq = Real_sequence\{([^}]*)\}", body).group(1)),
                       "mu_T": 1.0, "watched_site": int(re.search(r"m\.site == (\d+)", body).group(1)),
                       "expected_p_unusual": float(re.search(r"expected_p_unusual = ([0-9.]+);", body).group(1)),
                       "expected_p_super_unusual": float(re.search(r"expected_p_super_unusual = ([0-9.]+);", body).group(1)), "sigmas": 3}
    assert len(graft_tests) == 9 and len(peel_tests) == 3 and len(move_cases) == 5 and history is not None, (len(graft_tests), len(peel_tests), len(move_cases))
    return {"mu_JC": mu_JC, "states": "A C G T = 0 1 2 3; a mutation is [from, site, to, t], a missation [site, from], a delta [site, from, to]; node times of -1.797e308 / t_min, t_max of -+3.4e38 are the reference's -DBL_MAX / -+FLT_MAX",
            "fixtures": fixtures, "analyze_graft": graft_tests, "peel_apply": peel_tests, "full_spr_move": move_cases, "sample_mutational_history": history}


if __name__ == "__main__":
    pop, sk1 = pop_model_cases()
    iv, sk2 = interval_set_cases()
    sc = scalable_coalescent_case()
    out = {"source": "expectations of the reference's tests/pop_model_tests.cpp, interval_set_tests.cpp, scalable_coalescent_tests.cpp, phylo_tree_tests.cpp, spr_move_tests.cpp, phylo_tree_calc_tests.cpp, tree_editing_tests.cpp, evaluated by tests/golden/make_reference_expectations.py",
           "pop_model": pop, "interval_set": iv, "scalable_coalescent": sc, "phylo_tree_queries": phylo_tree_query_cases(), "spr_move": spr_move_cases(), "phylo_tree_calc": phylo_tree_calc_cases(), "tree_editing": tree_editing_cases(),
           "not_converted": {"pop_model_tests.cpp": sk1, "interval_set_tests.cpp": sk2, "why": "accessors, printing, iterator-identity and derivative expectations (off the hot path)"}}
    json.dump(out, open(OUT, "w"), indent=0)
    print("pop_model cases %d (skipped %d) | interval_set cases %d (skipped %d) | scalable_coalescent stages %d" % (len(pop), sk1, len(iv), sk2, len(sc["stages"])))
    print("spr_move: %d fixtures, %d analyze_graft tests, %d peel / apply tests, %d full_spr_move case lists" % (len(out["spr_move"]["fixtures"]), len(out["spr_move"]["analyze_graft"]), len(out["spr_move"]["peel_apply"]), len(out["spr_move"]["full_spr_move"])))
    from collections import Counter
    print(Counter(c["test"] for c in pop)); print(Counter(c["test"] for c in iv))
