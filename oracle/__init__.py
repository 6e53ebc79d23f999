"""oracle/ -- TEST INFRASTRUCTURE, not product code.

Two CPU checkers for the per-VFO IQ chain, behind one Python interface
(:class:`oracle.binding.OracleVfo`, method names = the reference's ``vfo`` setters):

* ``port``      -- oracle/liborc.so, the plain-C restatement (oracle/vfo_oracle.c).  Travels
                   to the GPU box; rebuilt by ``__graft_entry__.build()`` with gcc.
* ``reference`` -- oracle/_ref/libsdrref.so, the reference's own sources compiled unmodified
                   by oracle/ref/Makefile (only buildable where /root/reference exists).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this
package.  Nothing under sdrreceiver_amd/ does.
"""
