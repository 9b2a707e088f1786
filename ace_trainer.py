"""Drop-in for `from ace_trainer import TrainerACE` (train_ace.py:20,240-241 of the reference):

    trainer = TrainerACE(options)     # options: the argparse namespace of train_ace.py (same flags, acezero_amd.cli.train_parser)
    trainer.train()

The reference's class creates the dataset, the training buffer, the network and the optimiser in its constructor and runs the
epochs in train(); here both halves are acezero_amd.cli.train_with_options (buffer creation and the training loop on the GPU),
and the files it leaves behind are the reference's (<output>.pt fp16 head, <output>.txt log, poses_<id>_preliminary.txt)."""
from acezero_amd.cli import train_with_options


class TrainerACE:
    def __init__(self, options):
        if options.batch_size % 512 != 0:
            raise ValueError("batch_size must be a multiple of 512 (train_ace.py:138)")
        self.options = options
        self.iteration = 0
        self.training_start = None

    def train(self):
        import time
        self.training_start = time.time()
        return train_with_options(self.options)
