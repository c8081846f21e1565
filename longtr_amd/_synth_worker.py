"""A generator worker of synth.config_loci: one pickled job on stdin, the loci (pickled) on stdout."""
import pickle
import sys

from longtr_amd import synth

if __name__ == "__main__":
    sys.stdout.buffer.write(pickle.dumps(synth._gen_loci(pickle.loads(sys.stdin.buffer.read()))))
