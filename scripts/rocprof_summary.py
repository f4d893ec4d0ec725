"""Summarise a rocprofv3 rocpd database (bench_results.db) into the per-kernel table committed under
profiles/: calls, total/avg/min/max duration, share of GPU time (the --stats view of --kernel-trace)."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"| `{n[:120]}` | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.2f} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "a").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
