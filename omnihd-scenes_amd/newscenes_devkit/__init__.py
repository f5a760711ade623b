"""Evaluation slice of the reference's ``newscenes_devkit``: the detection
matching/AP/TP/NOS arithmetic under the same module paths and names.  The database loader
(newscenes.py), rendering and tracking evaluation are out of scope."""
