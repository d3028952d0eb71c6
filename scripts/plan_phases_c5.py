"""scripts/plan_phases.py on the C5 workload (300k Gaussians, 4K), four frames per step"""
import os, sys
os.environ["SOAR_ONLY4"] = "1"
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "plan_phases.py")).read().replace('"C3"', '"C5"')
exec(compile(src, "plan_phases_c5", "exec"))
