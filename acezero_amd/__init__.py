"""acezero_amd: MI355X-native hot path of ACE Zero behind the reference's call surfaces."""

# The 16-bit operand format of encoder and head when neither an argument (dtype= / --compute_dtype) nor $ACEZ_DTYPE names one.
# "bf16": BASELINE.json's north_star format; "fp16": the reference's autocast arithmetic. DESIGN.md section 0 holds the evidence for the choice.
DEFAULT_DTYPE = "bf16"
