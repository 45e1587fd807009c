"""Worker-process initialiser of elfi_amd.gpu_client (kept free of `import elfi` so that a spawned
worker can run a user-supplied setup hook before ELFI itself is imported)."""
import os


def init(gpu, setup):
    os.environ['ELFI_AMD_DEVICE'] = str(gpu)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if setup is not None:
        setup()
