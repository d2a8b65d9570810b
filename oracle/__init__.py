"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32) restatement of the Meta-Transformer encoder hot path and of the
Data2Seq tokenizers that feed it.  It is the *checker* for the HIP path in
``metatransformer_amd`` -- never the thing measured or shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import from here.  The product package (``metatransformer_amd``) must
never import ``oracle`` (tests/test_boundary.py enforces this by grepping).

Parity pinning status
---------------------
The reference ships NO tests, golden vectors or fixtures for this path
(SURVEY.md section 4 / 8c), and its arithmetic lives in un-vendored ``timm``
(pinned 0.4.12 / 0.4.5 / 0.8.0.dev0).  The reference does however carry a verbatim
in-tree copy of timm's Block ("borrowed from TIMM"):

    PointCloud/openpoints/models/layers/attention.py:12-58   (Attention, Block)
    PointCloud/openpoints/models/layers/mlp.py:11-35         (Mlp)

``oracle/ref_loader.py`` imports that file UNMODIFIED (in the build container,
where /root/reference exists) and ``oracle/make_golden.py`` uses it to generate
the fixtures in ``tests/golden/``.  So the oracle is pinned against *outputs of
the reference's own code run here* -- not against reference-side tests, which do
not exist.
"""
