import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
splats, cfg = pkg.scenes.make_config("C4")
eng = pkg.Engine(0)
eng.upload(splats)
for f in range(3):
    cam = pkg.camera.make_camera(1920, 1080, sh_order=3, frame=f)
    eng.render(cam)
tw = eng.debug_tile_work()          # [ty, tx, 2]
ls, le, pv = eng.debug_tile_lists()
st = eng.stats()
S, sx, sy = st["super_tile"], st["stiles_x"], st["stiles_y"]
L = (le - ls).reshape(sy, sx)
scanned = tw[..., 0].astype(np.int64); fetched = tw[..., 1].astype(np.int64)
ty, tx = scanned.shape
tot_pref_max = tot_pref_sat = 0
rows = []
for j in range(sy):
    for i in range(sx):
        blk = scanned[j*S:(j+1)*S, i*S:(i+1)*S]
        fb = fetched[j*S:(j+1)*S, i*S:(i+1)*S]
        n = L[j, i]
        sat = blk < n            # stopped before the end of the list
        m_all = blk.max() if blk.size else 0
        m_sat = blk[sat].max() if sat.any() else 0
        tot_pref_max += min(n, m_all*1.25+1024); tot_pref_sat += min(n, m_sat*1.25+1024)
        rows.append((j, i, n, int(m_all), int(m_sat), int(sat.sum()), blk.size, int(fb[~sat].sum())))
print("lists total", L.sum(), "prefix(max all)", int(tot_pref_max), "prefix(max saturated)", int(tot_pref_sat))
print("tiles", scanned.size, "non-saturated tiles", int(sum(r[6]-r[5] for r in rows)), "their gathers", int(sum(r[7] for r in rows)), "all gathers", int(fetched.sum()))
for r in rows[::9]:
    print(r)
q = np.quantile(scanned[scanned > 0], [0.5, 0.9, 0.99, 1.0]); print("scan depth quantiles", q)
