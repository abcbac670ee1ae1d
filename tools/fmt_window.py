import sys, json
for l in sys.stdin:
    if not l.startswith("{"): continue
    j = json.loads(l)
    if "window_markers" in j:
        print(j["window_markers"], j["ring_columns"], round(j["ms_per_flush"], 3), int(j["variants_per_s"]), round(j["int8_band_TOPs"]))
