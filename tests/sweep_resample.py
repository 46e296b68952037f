"""Sweep of the input-rate conversion (not a pytest module): 12 input rates x CBR / VBR / ABR / mono settings x
output rate left open or forced; wherever the compiled reference ends up at an MPEG-1 rate and resamples, the
resolved constants and the bytes of oracle/orc_resample.c + frame oracle + packer must equal the reference's;
wherever it ends up at an MPEG-2 / 2.5 rate the product must refuse.  Needs oracle/_ref.  CPU only."""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "deprecated-lame-mirror_amd"))
import helpers, lamehip, test_resample as T
import ctypes as C
oracle, reference = helpers.Oracle(), helpers.Reference()
ok = bad = skipped = 0
for rate_in in (8000, 11025, 16000, 22050, 24000, 32000, 37800, 44100, 48000, 64000, 88200, 96000):
    for kw in [dict(brate=b) for b in (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 256, 320)] + [dict(vbr_q=q) for q in range(10)] + [dict(abr=a) for a in (60, 90, 105, 120, 200)] + [dict(brate=b, channels=1) for b in (32, 48, 64, 96, 128)]:
        for out in (0, 32000, 44100, 48000):
            lib = reference.lib
            lib.refh_option.argtypes = [C.c_char_p, C.c_float]
            try:
                h = T.open_reference(reference, rate_in, kw, out)
            except AssertionError:
                skipped += 1; continue
            rcfg = T.LhConfig(); lib.refh_get_config(h, C.byref(rcfg)); lib.refh_close(h)
            if rcfg.samplerate not in (32000, 44100, 48000):
                # the product must refuse
                try:
                    e = T.open_product(rate_in, kw, out, False); print("ACCEPTED but MPEG-2", rate_in, kw, out); bad += 1; e.close()
                except AssertionError:
                    skipped += 1
                continue
            if not oracle.lib.orc_rs_needed(rate_in, rcfg.samplerate):
                skipped += 1; continue
            try:
                T.test_resampled_oracle_matches_reference(rate_in, kw, out, rcfg.samplerate, oracle, reference)
                ok += 1
            except AssertionError as ex:
                bad += 1; print("BAD", rate_in, kw, out, str(ex)[:200])
print("ok", ok, "bad", bad, "refused by both / mpeg-2", skipped)
