"""profiles/traffic.json from a parsed rocprofv3 PMC summary (tools/parse_profiles.py output).

HBM bytes per launch of the dominant kernel = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024:
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the bytes of a coalesced
streaming read (MI355X_MICROARCH.md, HBM section) -- on this kernel's 8-B-per-lane SoA rows the doubled value
reproduces the known byte count (176 B x points) to 0.02 %, which is the calibration that section asks for.
Usage: python tools/make_traffic_json.py gpurun_out/prof_TAG/summary_TAG.json <kernel substring> <points>"""
import json
import sys

summary, needle, points = sys.argv[1], sys.argv[2], int(sys.argv[3])
s = json.load(open(summary))


def pick(section, counter):
    for k, v in s[section].items():
        if needle in k:
            return k, v[counter]["mean"]
    raise SystemExit(f"kernel {needle} not in {section}")


name, fetch_kib = pick("pmc_fetch", "FETCH_SIZE")
_, write_kib = pick("pmc_write", "WRITE_SIZE")
read_b = fetch_kib * 1024 * 2
write_b = write_kib * 1024
out = {"kernel": name[:120], "points_per_launch": points,
       "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib, "fetch_correction": 2.0,
       "read_bytes": read_b, "write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
       "hbm_bytes_per_point": (read_b + write_b) / points, "source": summary}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
