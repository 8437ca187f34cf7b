"""Summarise rocprofv3 (rocpd sqlite) outputs of scripts/profile.sh into JSON: per-kernel durations and
per-launch PMC sums for k_run_moves.  Usage: summarize_prof.py <dir> [--launch-moves N]"""
import glob, json, os, sqlite3, sys
out = sys.argv[1]
res = {"kernel_ms": {}, "pmc_k_run_moves_per_launch": {}}


def tables(con):
    t = [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
    return t[0].replace("rocpd_kernel_dispatch", "") if t else None


for f in sorted(glob.glob(os.path.join(out, "*", "*.db"))):
    con = sqlite3.connect(f)
    sfx = tables(con)
    if sfx is None:
        continue
    names = {r[0]: r[1] for r in con.execute(f"select id, kernel_name from rocpd_info_kernel_symbol{sfx}")}
    disp = list(con.execute(f"select kernel_id, start, end, event_id, private_segment_size, group_segment_size, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch{sfx}"))
    tag = os.path.basename(os.path.dirname(f))
    if tag == "trace":
        ks = {}
        for kid, s, e, ev, priv, grp, gx, wx in disp:
            n = names.get(kid, "?")
            short = n.split("(")[0]
            ks.setdefault(short, []).append((e - s) / 1e6)
            res.setdefault("kernel_launch_info", {})[short] = {"scratch_bytes_per_lane": priv, "lds_bytes": grp, "grid": gx, "workgroup": wx}
        tot = sum(sum(v) for v in ks.values())
        res["kernel_ms"] = {k: {"calls": len(v), "total_ms": sum(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v), "pct": 100 * sum(v) / tot} for k, v in ks.items()}
    else:
        pmc_names = {r[0]: r[1] for r in con.execute(f"select id, name from rocpd_info_pmc{sfx}")}
        ev_kernel = {ev: names.get(kid, "?") for kid, s, e, ev, *_ in disp}
        acc = {}
        for ev, pid, val in con.execute(f"select event_id, pmc_id, value from rocpd_pmc_event{sfx}"):
            if "k_run_moves" not in ev_kernel.get(ev, ""):
                continue
            acc.setdefault(pmc_names.get(pid, str(pid)), {}).setdefault(ev, 0.0)
            acc[pmc_names.get(pid, str(pid))][ev] += val
        for name, per_ev in acc.items():
            res["pmc_k_run_moves_per_launch"][name] = sum(per_ev.values()) / max(1, len(per_ev))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
