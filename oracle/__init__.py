"""CPU oracle for the LARP tokenizer hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (video-tokenizer_amd/) must never import it.
"""
