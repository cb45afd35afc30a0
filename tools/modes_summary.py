"""Per dispatch of a `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format json` run: kernel name, duration and the counter's values PER
INSTANCE (the json output keeps one record per instance of the counter: XCD x channel for the TCC block), reduced to total / max / mean over
the instances and the imbalance max / mean.  Falls back to printing the shape of the file when the layout is not the expected one.
usage: modes_summary.py <dir> <COUNTER>"""
import glob
import json
import sys


def shape(o, depth=0, key=""):
    pad = "  " * depth
    if isinstance(o, dict):
        print("%s%s{%d keys}" % (pad, key, len(o)), file=sys.stderr)
        if depth < 6:
            for k, v in list(o.items())[:24]:
                shape(v, depth + 1, k + ": ")
    elif isinstance(o, list):
        print("%s%s[%d]" % (pad, key, len(o)), file=sys.stderr)
        if o and depth < 6:
            shape(o[0], depth + 1, "[0] ")
    else:
        print("%s%s%r" % (pad, key, o if not isinstance(o, str) else o[:60]), file=sys.stderr)


d, cname = sys.argv[1], sys.argv[2]
out = []
for f in sorted(glob.glob(d + "/**/*results.json", recursive=True)):
    J = json.load(open(f))
    root = J["rocprofiler-sdk-tool"][0] if "rocprofiler-sdk-tool" in J else J
    shape(root)
    try:
        names = {}
        for ks in root.get("kernel_symbols", []):
            names[ks.get("kernel_id")] = (ks.get("formatted_kernel_name") or ks.get("kernel_name") or "?").split("(")[0]
        disp = {}
        for r in root.get("buffer_records", {}).get("kernel_dispatch", []):
            di = r.get("dispatch_info", r)
            disp[di.get("dispatch_id")] = (names.get(di.get("kernel_id"), "?"), r.get("end_timestamp", 0) - r.get("start_timestamp", 0))
        for r in root.get("callback_records", {}).get("counter_collection", []):
            di = r.get("dispatch_data", {}).get("dispatch_info", {})
            vals = [x.get("value", x.get("counter_value")) for x in r.get("records", [])]
            vals = [v for v in vals if v is not None]
            if not vals:
                continue
            nm, dur = disp.get(di.get("dispatch_id"), (names.get(di.get("kernel_id"), "?"), 0))
            out.append(dict(kernel=nm, dispatch=di.get("dispatch_id"), dur_ns=dur, instances=len(vals), total=sum(vals), max=max(vals), min=min(vals),
                            imbalance=max(vals) / (sum(vals) / len(vals)) if sum(vals) else None,
                            per_instance=vals if len(vals) <= 256 else None))
    except Exception as e:  # the shape printed above tells how to read the file
        print("parse error: %r" % (e,), file=sys.stderr)
print(json.dumps(out))
