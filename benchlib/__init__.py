"""Measurement tooling of bench.py and scripts/ (NOT part of the product package elfi_amd/): the BOLFI legs of the bench line
and the timers that sit on the calls the reference's loops make into the device objects."""
