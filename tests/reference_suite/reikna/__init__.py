"""TEST INFRASTRUCTURE: the reference's test_gates.py does `from reikna.cluda import cuda_id`; nothing else of Reikna exists here."""
