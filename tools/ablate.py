import os, sys, subprocess, json
for stop in (1,2,3,4,5,0):
    env=dict(os.environ, ADSB_DEBUG_STOP=str(stop))
    out=subprocess.run([sys.executable,'bench.py','--steps','10','--warmup','3','--no-cpu-baseline','--buffers','2'],env=env,capture_output=True,text=True).stdout.strip().splitlines()[-1]
    r=json.loads(out)
    print(stop, r['roofline']['kernel_avg_ms'], r['ms_per_step'], r['device_stats_last_step'])
