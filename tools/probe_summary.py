import json
r=json.load(open("gpurun_out/e2e_cohort_probe.json"))
for row in r["rows"]:
    print(row["variant"][:60], row.get("same_vcf_as_first"), row.get("steady_seconds_per_segment"), row.get("warmed_seconds_per_segment"), row.get("cpu_seconds"), row.get("cpu_system_seconds"), row.get("warmed_cpu_seconds_per_segment"))
