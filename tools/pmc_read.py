#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc sqlite outputs: mean counter value and duration per kernel-name substring.
    python tools/pmc_read.py <dir-with-*/..._results.db> <kernel substring>"""
import glob, sqlite3, sys, collections
root, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(root + '/**/*_results.db', recursive=True)):
    con = sqlite3.connect(f)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    T = lambda key: next(t for t in tabs if t.startswith('rocpd_' + key))
    ks = {r[0]: r[1] for r in con.execute(f"select id, kernel_name from {T('info_kernel_symbol')}")}
    pm = {r[0]: r[1] for r in con.execute(f"select id, name from {T('info_pmc')}")}
    disp = {r[0]: (r[1], r[2], r[3]) for r in con.execute(f"select event_id, kernel_id, start, end from {T('kernel_dispatch')}")}
    acc = collections.defaultdict(list); dur = []
    for ev, pid, val in con.execute(f"select event_id, pmc_id, value from {T('pmc_event')}"):
        if ev in disp and pat in ks.get(disp[ev][0], ''):
            acc[pm[pid]].append(val)
    for ev, (kid, st, en) in disp.items():
        if pat in ks.get(kid, ''):
            dur.append(en - st)
    print(f.split('/')[-2], 'dispatches', len(dur), 'mean ns', sum(dur) / max(1, len(dur)))
    for k, v in acc.items():
        print('   ', k, 'n=%d mean=%.5g' % (len(v), sum(v) / len(v)))
