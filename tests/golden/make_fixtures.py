"""Regenerates the fixtures in this directory from the reference's own DATA files (run in the build container, where
/root/reference exists; the GPU box only sees the committed outputs).

  nnet.proto            <- google/nnet.proto: the recipe's network prototype (configuration data, 6 lines): the
                           <LstmProjectedStreams> line is what InitData (...streams.h:55-99) parses
  model_headers.txt     <- the component header lines of the two example nnet files in README.md (marker, output-dim,
                           input-dim, then the WriteData tokens, ...streams.h:133-150): what a model written by the
                           reference looks like on disk, up to the first matrix bracket

  feature_transform.nnet.txt <- google/feature_transform.nnet.txt: a model file WRITTEN BY KALDI (text mode, two
                           vector-parameter components, 860 bytes): a genuine sample of the on-disk token / vector
                           encoding for the reader in include/klstm_kaldi_io.hpp

None is source code; no arithmetic is pinned by them (the reference ships no numeric vectors) --
they pin the FORMAT side of the drop-in boundary (SURVEY 8(b) "file format") to reference-provided data."""
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

shutil.copy(os.path.join(REF, "google", "nnet.proto"), os.path.join(HERE, "nnet.proto"))
shutil.copy(os.path.join(REF, "google", "feature_transform.nnet.txt"), os.path.join(HERE, "feature_transform.nnet.txt"))
lines = []
for line in open(os.path.join(REF, "README.md")):
    m = re.match(r"^(<(?:LstmProjectedStreams|LstmProjected|TimeShift|Transmit)>[^\[]*?)\s*(\[ \.\.\.)?\s*$", line)
    if m:
        lines.append(m.group(1).rstrip())
open(os.path.join(HERE, "model_headers.txt"), "w").write("\n".join(lines) + "\n")
print(open(os.path.join(HERE, "model_headers.txt")).read())
