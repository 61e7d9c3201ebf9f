"""CPU oracle of the demod_2400 path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Nothing under dump1090_rs_amd/ does.
"""
