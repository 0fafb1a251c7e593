"""Public API (mirror of xanthos/model.py:21-132): ``Xanthos(ini).execute(args)`` and ``run_model(ini)``."""
import argparse
import logging
import os
import sys

from . import launch
from .configurations import ConfigRunner
from .ini_reader import ConfigReader


class Xanthos:
    """The pm_abcd_mrtm configuration of Xanthos on MI355X."""

    def __init__(self, ini):
        self.ini = ini
        self.config = None

    def stage(self, mem_args):
        self.config = ConfigReader(self.ini)
        self.config.update(mem_args)
        rank, local_rank, world = launch.env_world()
        if world > 1:                      # one process per GPU: this rank's device (XH_ONE_DEVICE=1: all on GPU 0, test boxes)
            self.config.device = 0 if os.environ.get('XH_ONE_DEVICE') == '1' else local_rank
        os.makedirs(self.config.OutputFolder, exist_ok=True)
        logger = logging.getLogger()
        logger.setLevel(logging.INFO)
        self._handlers = [logging.StreamHandler(sys.stdout)]
        if rank == 0:                      # (the log file belongs to the rank that writes the outputs)
            self._handlers.append(logging.FileHandler(os.path.join(self.config.OutputFolder, 'logfile.log')))
        for h in self._handlers:
            h.setFormatter(logging.Formatter('%(levelname)s: %(message)s'))
            logger.addHandler(h)

    def execute(self, args={}):
        """Run the configuration; ``args`` overrides settings in memory (model.py:82-98). Returns the Components."""
        self.stage(args)
        try:
            return ConfigRunner(self.config).run()
        finally:
            self.cleanup()

    def cleanup(self):
        logging.info('End of {0}'.format(self.config.ProjectName))
        logger = logging.getLogger()
        for h in getattr(self, '_handlers', []):
            logger.removeHandler(h)
            h.close()


def run_model(config_file, gpus=None):
    """Run Xanthos from a configuration file (model.py:111-121).

    ``gpus`` > 1 (or XH_GPUS in the environment) in a process that no launcher started: the 235 basins are sharded over that
    many GPUs of this node -- ``gpus`` rank processes are started (children of this one, before anything here touches a
    GPU), each runs this same function as one rank, rank 0 gathers and writes the outputs; returns None (the results are
    the files).  Under a launcher (RANK / WORLD_SIZE set) the call IS one rank."""
    if gpus is None and os.environ.get('XH_GPUS'):
        gpus = int(os.environ['XH_GPUS'])
    if gpus and int(gpus) > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        env = dict(os.environ)             # the rank processes import this very package, wherever the caller found it
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env['PYTHONPATH'] = pkg_parent + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
        rc = launch.spawn(int(gpus), ['-m', 'xanthos_amd.model', os.path.abspath(config_file)], env=env,
                          one_device=os.environ.get('XH_ONE_DEVICE') == '1')
        if rc != 0:
            raise RuntimeError('run_model on {} GPUs: a rank exited with code {}'.format(gpus, rc))
        return None
    try:
        return Xanthos(config_file).execute()
    finally:
        launch.close_group()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('config_file', type=str, help='Full path with file name to INI configuration file.')
    parser.add_argument('--gpus', type=int, default=None, help='GPUs of this node to shard the basins over (one process each)')
    a = parser.parse_args()
    run_model(a.config_file, gpus=a.gpus)
