"""On-disk formats either side of the stress-update path: the material-point subset of
the reference's YAML deck, the deformation-history / QoI-data readers and the result writers.  File names,
array shapes and JSON keys are the reference's (`cmad/io/*.py`) so that outputs are interchangeable."""
