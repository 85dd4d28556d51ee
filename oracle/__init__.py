"""
oracle/ -- CPU restatement of the mycroft-precise hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import anything from this package, and there only as the *checker*: nothing under
``mycroft_precise_amd/`` imports it, and the product path fails loudly when the HIP
library is missing instead of falling back to this code.

What is restated, and how firmly each piece is pinned
-----------------------------------------------------
* Streaming glue (``Listener.update_vectors`` / ``update``, ``vectorize``,
  ``add_deltas``, ``buffer_to_audio``, ``ThresholdDecoder``) -- restated from
  ``/root/reference/precise/{network_runner,vectorization,util,threshold_decoder,
  functions,params}.py``.  PINNED: ``oracle/gen_golden.py`` runs the reference's own,
  unmodified ``precise.network_runner.Listener`` / ``ThresholdDecoder`` in the build
  container and commits its outputs under ``tests/golden/``; ``tests/test_oracle.py``
  checks this restatement against those fixtures.
* MFCC arithmetic -- third-party **sonopy 0.1.2** (``requirements.txt:35``), NOT vendored
  in the reference and not installable offline.  ``oracle/sonopy_restated.py`` restates
  its published algorithm.  **PARITY UNPINNED** against real sonopy bits.
* GRU / Dense arithmetic -- third-party **Keras 2.2.4 / TensorFlow 1.13.1**
  (``requirements.txt:11,38``), not vendored, not installable.  ``oracle/keras_gru.py``
  restates ``GRUCell.call`` (implementation 1, reset_after=False, hard_sigmoid) and the
  Dense+sigmoid head.  **PARITY UNPINNED** against real Keras/TF bits.

So: the golden fixtures pin *the reference's own code* (everything the reference repo
actually owns on this path) with the restated third-party arithmetic plugged into its
documented seams (``sonopy`` module, ``runner_cls``).
"""
