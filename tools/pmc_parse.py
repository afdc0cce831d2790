#!/usr/bin/env python3
"""Per-dispatch counter values from a rocprofv3 --pmc rocpd database.
Usage: tools/pmc_parse.py results.db [kernel-name-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection "
                      "order by dispatch_id").fetchall()
    by = {}
    for did, name, cn, v, dur in rows:
        if pat in name:
            by.setdefault((did, name, dur), {})[cn] = by.get((did, name, dur), {}).get(cn, 0) + v
    for (did, name, dur), c in sorted(by.items()):
        print(f"dispatch {did:5d} {dur/1e3:9.2f} us  " + "  ".join(f"{k}={v:.0f}" for k, v in sorted(c.items())) + f"  {name[:60]}")


if __name__ == "__main__":
    main()
