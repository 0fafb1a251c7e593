"""Public API (mirror of xanthos/model.py:21-132): ``Xanthos(ini).execute(args)`` and ``run_model(ini)``."""
import argparse
import logging
import os
import sys

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
        os.makedirs(self.config.OutputFolder, exist_ok=True)
        logger = logging.getLogger()
        logger.setLevel(logging.INFO)
        self._handlers = [logging.StreamHandler(sys.stdout),
                          logging.FileHandler(os.path.join(self.config.OutputFolder, 'logfile.log'))]
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


def run_model(config_file):
    """Run Xanthos from a configuration file (model.py:111-121)."""
    return Xanthos(config_file).execute()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('config_file', type=str, help='Full path with file name to INI configuration file.')
    run_model(parser.parse_args().config_file)
