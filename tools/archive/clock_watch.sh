#!/bin/bash
# Sample GPU clock levels / power / temperatures from sysfs while a command runs: is the run-to-run spread of the
# headline bench a power-management state?   usage: clock_watch.sh <out.txt> -- <command...>
out=$1; shift; shift
( while true; do
    t=$(date +%s.%N | cut -c1-14)
    line="$t"
    for dev in /sys/class/drm/card*/device; do            # several cards may be visible: log each, the busy one stands out
      hw=$(ls -d $dev/hwmon/hwmon* 2>/dev/null | head -1)
      [ -z "$hw" ] && continue
      s=$(grep '\*' $dev/pp_dpm_sclk 2>/dev/null | tr -d '\n')
      p=$(cat $hw/power1_average 2>/dev/null || cat $hw/power1_input 2>/dev/null)
      line="$line | sclk[$s] power_uW=$p"
    done
    echo "$line"
    sleep 0.2
  done ) > $out 2>/dev/null &
wp=$!
"$@"
kill $wp
