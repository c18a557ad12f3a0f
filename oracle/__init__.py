"""Oracle: CPU restatements used ONLY as the checker (tests/, smoke(), bench cpu_baseline)."""
