"""oracle/ — CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (med-ts-llm_amd/) never imports it and fails loudly without its HIP library.
"""
