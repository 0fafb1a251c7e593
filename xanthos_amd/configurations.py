"""ConfigRunner (mirror of xanthos/configurations.py:17-141): validates the selectors and runs the components."""
import logging

from .components import Components


class ConfigRunner:
    PET_COMPONENTS = ['pm']                 # the reference also lists hs / hargreaves / thornthwaite (:61)
    RUNOFF_COMPONENTS = ['abcd']            # reference: + gwam (:62)
    ROUTING_COMPONENTS = ['mrtm']           # (:63)

    def __init__(self, config):
        self.run_pet = config.pet_module in self.PET_COMPONENTS
        self.run_runoff = config.runoff_module in self.RUNOFF_COMPONENTS
        self.run_routing = config.routing_module in self.ROUTING_COMPONENTS
        # pm / abcd / mrtm iterate internally: all *_timestep are 0 and no whole-model spin-up (:69-85)
        self.pet_timestep = self.runoff_timestep = self.routing_timestep = 0
        self.spinup = False
        self.config = config

    def run(self):
        if not (self.run_pet or self.run_runoff or self.run_routing):
            logging.warning('Selected configuration {0} not supported.'.format(self.config.mod_cfg))
            return None
        import time
        c = Components(self.config)
        c.simulation(run_pet=self.run_pet, run_runoff=self.run_runoff, run_routing=self.run_routing,
                     pet_num_steps=0, runoff_num_steps=0, routing_num_steps=0, notify='Simulation')
        t = time.time()
        c.accessible_water()          # post-processors, then the outputs: the reference's order (configurations.py:117-136)
        c.drought()
        c.timings['post'] = time.time() - t
        t = time.time()
        c.output_simulation()
        c.timings['write'] = time.time() - t
        logging.info('run_model phases (s): ' + ', '.join('{} {:.3f}'.format(k, v) for k, v in c.timings.items()))
        return c
