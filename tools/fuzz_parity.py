#!/usr/bin/env python3
"""Randomised parity fuzz on a GPU box: random sizes (1..420 x 1..700), every Farneback parameter (polyN 1-7, winSize
2-65, pyrLevels 0-6, iterations 0-5, pyrScale 0.3-0.9, box / Gaussian window, initial-flow flag, polySigma incl. 0),
four image kinds, random span / threshold — dense flow and vector list of the engine must equal the oracle's bit for
bit.    tools/fuzz_parity.py [seed]      (3 000 cases or 240 s, whichever comes first)
Round 2: seeds 2026, 7 and 99 = 6 400 cases, 0 mismatches.  FUZZ_FOCUS=win50: winSize 50 / 51 on 481..1500-pixel-wide images.
FUZZ_FOCUS=mfree (round 5): tw_flow_iter — winSize 30 / 31, Gaussian window, 320..1500-pixel-wide images of 20..560 rows, forced
for single pairs (TW_MFREE=2, no single-pair stream split) so that every level of >= 320 columns runs it.
FUZZ_FOCUS=twin (round 5): the single-pair schedule of twin launches — sizes that are multiples of 8 (256..1280 x 200..800),
polyN 7, winSize 30 / 31, pyrScale 0.5, pyrLevels 3-5, 1-5 iterations, slots = 1.
FUZZ_FOCUS=big (round 6): single pairs of 1700..2300 x 900..1300 pixels (any remainder), whose level 0 takes the 224 x 8 tiles of
tw_blur_solve4q — polyN 5 / 7, winSize 30 / 31, pyrScale 0.5 / 0.6, 1-4 levels, 1-3 iterations (a case takes ~3 s of oracle time).
FUZZ_FOCUS=ramp (round 6): the cold-start ramp — a 64-slot engine, 17..64 HOST pairs (two distinct ones, alternating) of 40..420 x 40..700
pixels submitted into an idle stream, any parameters: every pair's vectors at threshold 0 / 0.5 against the oracle's, and the launch
counters must show the pieces (more than one polyexp launch per level) whenever more than 16 pairs were submitted."""
import sys, os, time
if os.environ.get("FUZZ_FOCUS","")=="mfree":
    os.environ["TW_MFREE"]="2"; os.environ["TW_LATENCY_STREAMS"]="0"
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,os.path.join(R,"tidal-wave_amd")); sys.path.insert(0,os.path.join(R,"oracle"))
import numpy as np, twflow as T, oracle as O
O.build(); O.lib()
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 2026)
bad=0; n=0; unsupported=0; t0=time.time(); fi_total=[0]
while n < 3000 and time.time()-t0 < float(os.environ.get("FUZZ_SECONDS","240")):
    h,w=int(rng.integers(1,420)),int(rng.integers(1,700))
    focus=os.environ.get("FUZZ_FOCUS","")  # "win50": the 51-tap window kernels on wide levels (round 3's new default)
    if focus=="win50": h,w=int(rng.integers(1,520)),int(rng.integers(481,1500))
    if focus=="mfree": h,w=int(rng.integers(20,560)),int(rng.integers(320,1500))
    if focus=="twin": h,w=8*int(rng.integers(25,101)),8*int(rng.integers(32,161))
    if focus=="big": h,w=int(rng.integers(900,1300)),int(rng.integers(1700,2300))  # round 6: a single pair's 224 x 8-tile level 0 (tw_blur_solve4q)
    kw=dict(polyN=int(rng.integers(1,8)), winSize=int(rng.choice([50,51])) if focus=="win50" else int(rng.choice([30,31])) if focus=="mfree" else int(rng.integers(2,66)), pyrLevels=int(rng.integers(0,7)),
            pyrIterations=int(rng.integers(0,6)), pyrScale=float(rng.choice([0.3,0.45,0.5,0.55,0.6,0.7,0.75,0.8,0.9])),
            flags=int(rng.choice([256,260])) if focus=="mfree" else int(rng.choice([0,256,4,260])), polySigma=float(rng.choice([0.0,0.8,1.1,1.5,2.2])))
    if focus=="big":
        kw=dict(polyN=int(rng.choice([5,7])), winSize=int(rng.choice([30,31])), pyrLevels=int(rng.integers(1,5)), pyrIterations=int(rng.integers(1,4)),
                pyrScale=float(rng.choice([0.5,0.6])), flags=int(rng.choice([256,260])), polySigma=float(rng.choice([1.1,1.5])))
    if focus=="twin":
        kw=dict(polyN=7, winSize=int(rng.choice([30,31])), pyrLevels=int(rng.integers(3,6)), pyrIterations=int(rng.integers(1,6)),
                pyrScale=0.5, flags=int(rng.choice([256,260])), polySigma=float(rng.choice([1.1,1.5])))
    kind=int(rng.integers(0,4))
    a=rng.integers(0,256,(h,w),dtype=np.uint8)
    if kind==0: a=(a//64*64).astype(np.uint8)
    if kind==1: a[:]=int(rng.integers(0,256))
    b=np.roll(a,int(rng.integers(-3,4)),axis=int(rng.integers(0,2))).copy()
    if kind==2 and h>8 and w>8: b[h//3:h//2,w//4:w//2]=0
    span=int(rng.integers(1,15)); thr=float(rng.choice([0.0,0.5,2.0,5.0]))
    if focus=="ramp":
        if os.environ.get("FUZZ_RAMP_MFREE"):  # the pieces through tw_flow_iter: default window, 320..900 columns
            h,w=int(rng.integers(60,420)),int(rng.integers(320,900))
            kw.update(winSize=int(rng.choice([30,31])), flags=int(rng.choice([256,260])), polyN=int(rng.choice([5,7])), pyrIterations=int(rng.integers(1,6)))
            a=rng.integers(0,256,(h,w),dtype=np.uint8); b=np.roll(a,int(rng.integers(-3,4)),axis=int(rng.integers(0,2))).copy()
        if h<40 or w<40: continue
        thr=float(rng.choice([0.0,0.5])); npairs=int(rng.integers(17,65))
        a2=np.ascontiguousarray(a[::-1]); b2=np.roll(a2,2,axis=1).copy()
        pairs=[(a,b),(a2,b2)]
        try:
            with T.Engine(0,T.default_params(**kw),slots=64) as e:
                e.launch_counts(reset=True)
                tk=[e.submit(*pairs[i%2],span,thr) for i in range(npairs)]
                got=[e.wait(t)["vector"] for t in tk]
                cnt=e.launch_counts(); lv=e.num_levels(w,h); fi_total[0]+=cnt["tw_flow_iter"]+cnt["tw_flow_iter_ups"]
        except T.TwError as ex:
            if ex.code==T.TW_E_UNSUPPORTED: unsupported+=1; continue
            raise
        want=[]
        for x,y in pairs:
            wx,wy=O.farneback(x,y,O.default_params(**kw)); want.append(O.span_scan(wx,wy,span,thr))
        pieces=cnt["tw_polyexp"]//(lv+1) if cnt["tw_polyexp"]%(lv+1)==0 else -1
        ok=all(g==want[i%2] for i,g in enumerate(got)) and (cnt["tw_polyexp"]==0 or pieces==(3 if npairs>32 else 2))
        n+=1
        if not ok:
            bad+=1; print("MISMATCH",h,w,kw,span,thr,kind,npairs,pieces,dict(cnt), flush=True)
        continue
    try:
        with T.Engine(0,T.default_params(**kw),slots=1 if focus in ("twin","big") else 2) as e:
            gx,gy,_=e.calculate_internal(a,b)
            v=e.diff(a,b,span,thr)["vector"]
    except T.TwError as ex:
        if ex.code==T.TW_E_UNSUPPORTED: unsupported+=1; continue
        raise
    wx,wy=O.farneback(a,b,O.default_params(**kw))
    ok=np.array_equal(gx,wx) and np.array_equal(gy,wy) and v==O.span_scan(wx,wy,span,thr)
    n+=1
    if not ok:
        bad+=1; print("MISMATCH",h,w,kw,span,thr,kind, float(np.nanmax(np.abs(gx-wx))), flush=True)
print("fuzz: %d cases, %d mismatches, %d unsupported parameter sets, %.0fs"%(n,bad,unsupported,time.time()-t0)+(" (tw_flow_iter launches: %d)"%fi_total[0] if fi_total[0] else ""))
