P='import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(v.get("proofs_per_s",0),2),round(v.get("single_proof_ms",0),1)) if "error" not in v else v for k,v in d.get("size_classes",{}).items()}, d.get("commit_2p26",{}).get("ms_per_commit"), round(d["value"],1), d.get("h2d_inclusive_proofs_per_s"))'
echo "default"; (time python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P") 2>&1 | grep -v "^$\|user\|sys"
echo "default again"; python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
