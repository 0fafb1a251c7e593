"""
oracle/ -- CPU restatement of the Xanthos PET -> runoff -> routing hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker
(or as the timed CPU baseline) -- never as a fallback for the HIP path.  The
product package ``xanthos_amd`` never imports ``oracle``; it raises if the HIP
library is missing.

What is restated (reference = JGCRI/xanthos v2.4.1, pure numpy):

* ``oracle.pm``    <- xanthos/pet/penman_monteith.py:17-477
* ``oracle.abcd``  <- xanthos/runoff/abcd.py:41-422
* ``oracle.mrtm``  <- xanthos/routing/mrtm.py:16-258, xanthos/components.py:249-296
* ``oracle.calib`` <- xanthos/calibrate/calibrate_abcd.py:134-213
* ``oracle.months``<- xanthos/utils/general.py:15-50

Parity pinning: the reference ships no runnable golden vectors for this path
(its only regression test needs a Zenodo archive that is absent).  The oracle
is therefore pinned against outputs of the reference itself, imported in the
build container by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (inputs + reference outputs).  ``tests/test_oracle_golden.py``
checks every function here against those vectors.
"""
