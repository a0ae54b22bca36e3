"""Developer script: one line per bench log under a gpurun_out/<tag>/ folder."""
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.log"))):
    for ln in open(f):
        if ln.startswith('{"metric"'):
            d = json.loads(ln); c = d["config"]; ks = d["roofline"]["kernels"]
            print(f"{os.path.basename(f):14s} {d['value']:8.0f} r-steps/s  {d['ms_per_step']:7.2f} ms/step  B={c['rollouts_per_gpu']:3d} K={c['workgroups_per_rollout']} N={c['workload'].split('N=')[1].split(',')[0]} "
                  f"self={c['mean_self_contacts_per_step']:.0f} pd={c['mean_pd_iters_per_step']:.1f} cg={c['mean_cg_iters_per_pd_iter']:.1f} adj={c['mean_adjoint_iters_per_step']:.1f} "
                  f"fwd={ks[0]['ms_per_step']:.2f} bwd={ks[1]['ms_per_step']:.2f} launches={ks[0]['launches']}")
